#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_kernels.py -q -x -k "conv2d or kernel_names or lstm" -p no:cacheprovider 2>&1 | tail -3
for V in "JAF_WGRAD_FAST=1" "JAF_WGRAD_FAST=0" "JAF_WGRAD_FAST=1" "JAF_WGRAD_FAST=0"; do bash profiles/experiments/ab_w.sh wf_$V $V 2>&1 | head -16; done
