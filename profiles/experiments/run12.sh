#!/bin/bash
cd $GRAFT_REPO_ROOT
export JAFPRO_HIP_LIB=$PWD/scratch/x/lib_dmax.so JAF_CD_XG=1
for L in crn256 crn512_64 lstm1 dec4; do
  echo "== $L base: $(JAF_CD_X=0 python scratch/mb_conv.py bf16 $L 10 2>/dev/null | grep fwd)"
  for sl in 1 2 4 8 16; do for sb in 1024 100000000; do
  echo "== $L sleep=$sl blocks=$sb: $(JAF_CD_X=2048 JAF_CD_SLEEP=$sl JAF_CD_SLEEP_BLOCKS=$sb python scratch/mb_conv.py bf16 $L 10 2>/dev/null | grep fwd)"
  done; done
done
