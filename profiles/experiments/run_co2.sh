#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_kernels.py -q -x -k "conv2d or kernel_names or lstm" -p no:cacheprovider 2>&1 | tail -3
for V in "JAF_X=1" "JAFPRO_HIP_LIB=scratch/x/lib_prev2.so" "JAF_X=1" "JAFPRO_HIP_LIB=scratch/x/lib_prev2.so"; do bash profiles/experiments/ab_w.sh wg_${V%%=*} $V 2>&1 | head -18; done
