#!/bin/bash
export PYTHONPATH=$GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
for LIB in scratch/x/lib_base.so jafpro_amd/libjafpro_hip.so; do
 for PEN in 1.25 1.0; do
  echo "== $LIB JAF_PLAN_PEN2=$PEN"
  for L in crn256 crn259 crn512_64 vgg256_64 vgg64 crn32b dec4 enc3 inp36 lstm50; do
    JAFPRO_HIP_LIB=$GRAFT_REPO_ROOT/$LIB JAF_PLAN_PEN2=$PEN python profiles/experiments/mb_conv.py bf16 $L 20 2>&1 | grep fwd
  done
 done
done
echo "== correctness"
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "conv" 2>&1 | tail -3
