#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_modules.py tests/test_gpu_step_parity.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error" | tail -3
for i in 1 2; do
JAF_PLAN_PEN2=1.0 bash scratch/ab.sh x26_old$i | grep -E "ms/step|conv_dma_kernel<4, 4"
bash scratch/ab.sh x26_new$i | grep -E "ms/step|conv_dma_kernel<4, 4"
done
