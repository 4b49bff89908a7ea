#!/bin/bash
export PYTHONPATH=$GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
for RS in 0 1; do
 for PEN in 1.25 1.0; do
  echo "== JAF_CONV_RS=$RS JAF_PLAN_PEN2=$PEN"
  for L in crn256 crn259 crn512_64 vgg256_64 vgg64 crn32b; do
    JAF_CONV_RS=$RS JAF_PLAN_PEN2=$PEN python profiles/experiments/mb_conv.py bf16 $L 20 2>&1 | grep fwd
  done
 done
done
echo "== correctness with RS=1"
JAF_CONV_RS=1 JAF_PLAN_PEN2=1.0 timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "conv" 2>&1 | tail -5
