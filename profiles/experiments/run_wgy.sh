#!/bin/bash
export PYTHONPATH=$GRAFT_REPO_ROOT
export JAFPRO_HIP_LIB=$GRAFT_REPO_ROOT/scratch/x/lib_wgy.so
for L in crn256 lstm1; do
for X in 0 1 2 4 16 32 48 49; do
  echo "== $L JAF_WG_X=$X"
  JAF_WG_X=$X python profiles/experiments/mb_wgrad.py x $L 5 2>&1 | grep -i wgrad
done
done
