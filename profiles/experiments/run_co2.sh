#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_step.py tests/test_gpu_step_parity.py -q -x -p no:cacheprovider 2>&1 | tail -2
for V in 1 0 1 0 1 0; do
  JAF_EARLY_ADAM=$V python bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-config2 --parity-mode-steps 0 --no-roofline > gpurun_out/ea.json 2> gpurun_out/ea.err
  python -c "
import json; j=json.load(open('gpurun_out/ea.json')); print('JAF_EARLY_ADAM=$V: %.2f ms/step (median %.2f)' % (j['ms_per_step'], j['median_ms_per_step']))"
done
