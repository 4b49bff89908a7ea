#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_modules.py tests/test_gpu_step_parity.py -q -x -p no:cacheprovider 2>&1 | tail -2
for V in "JAF_X=1" "JAFPRO_HIP_LIB=scratch/x/lib_head2.so" "JAF_X=1" "JAFPRO_HIP_LIB=scratch/x/lib_head2.so" "JAF_X=1" "JAFPRO_HIP_LIB=scratch/x/lib_head2.so"; do
  env $V python bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-config2 --parity-mode-steps 0 > gpurun_out/bw.json 2> gpurun_out/bw.err
  python -c "
import json; j=json.load(open('gpurun_out/bw.json')); h=j['roofline'].get('hbm_kernels',{}); print('$V: %.2f ms/step (median %.2f)' % (j['ms_per_step'], j['median_ms_per_step']))"
done
