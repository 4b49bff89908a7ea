#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_step.py tests/test_gpu_step_parity.py -q -x -p no:cacheprovider > gpurun_out/t.log 2>&1; grep "passed\|failed" gpurun_out/t.log | tail -1
for V in 6 0 6 0 6 0; do
  JAF_REPACK_HEAD=$V python bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-config2 --parity-mode-steps 0 --no-roofline > gpurun_out/rh.json 2> gpurun_out/rh.err
  python -c "
import json; j=json.load(open('gpurun_out/rh.json')); print('JAF_REPACK_HEAD=$V: %.2f ms/step (median %.2f)' % (j['ms_per_step'], j['median_ms_per_step']))"
done
