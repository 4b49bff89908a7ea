#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_kernels.py -q -x -k "conv2d" -p no:cacheprovider 2>&1 | tail -2
for V in "JAF_X=1" "JAFPRO_HIP_LIB=scratch/x/lib_head3.so" "JAF_X=1" "JAFPRO_HIP_LIB=scratch/x/lib_head3.so"; do bash profiles/experiments/ab_w.sh p5_${V%%=*} $V 2>&1 | grep "ms/step\|<1, 5"; done
