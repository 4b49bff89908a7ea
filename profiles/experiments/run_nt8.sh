#!/bin/bash
export PYTHONPATH=$GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
for M in 0 2; do
  echo "== JAF_PLAN_NT8=$M"
  for L in crn256 crn259 crn512_64 vgg256_64 vgg64 crn32b crn64 vgg512_32; do
    JAF_PLAN_NT8=$M python profiles/experiments/mb_conv.py bf16 $L 20 2>&1 | grep fwd
  done
done
echo "== correctness with NT8=2"
JAF_PLAN_NT8=2 timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "conv" 2>&1 | tail -5
