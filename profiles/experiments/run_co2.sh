#!/bin/bash
cd $GRAFT_REPO_ROOT
JAF_WGRAD_RING=1 python -m pytest tests/test_gpu_kernels.py -q -x -k "conv2d" -p no:cacheprovider 2>&1 | tail -2
for V in "JAF_WGRAD_RING=1" "JAF_WGRAD_RING=0" "JAF_WGRAD_RING=1" "JAF_WGRAD_RING=0"; do bash profiles/experiments/ab_w.sh ring_${V##*=} $V 2>&1 | head -5; done
