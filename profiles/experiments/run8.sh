#!/bin/bash
cd $GRAFT_REPO_ROOT
export JAFPRO_HIP_LIB=$PWD/scratch/x/lib_dmax.so JAF_CD_XG=1
for L in crn256 crn512_64 vgg64; do
 for xf in 0 1 2 3 4 8 7 11; do
  echo "== $L X=$xf: $(JAF_CD_X=$xf python scratch/mb_conv.py bf16 $L 10 2>/dev/null | grep fwd)"
 done
done
