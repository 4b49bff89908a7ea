#!/bin/bash
cd $GRAFT_REPO_ROOT
export JAFPRO_HIP_LIB=$PWD/scratch/x/lib_wgx.so JAF_WGRAD_NO_DB=1
for L in lstm1 crn256 enc; do
 for xf in 0 1 2 4 3 9 5; do
  echo "== $L JAF_WG_X=$xf"; JAF_WG_X=$xf python scratch/mb_wgrad.py bf16 $L 5 2>/dev/null | grep wgrad
 done
done
