#!/bin/bash
cd $GRAFT_REPO_ROOT
for V in d tail dtail d tail dtail; do
  JAF_PREP_AT=$V python bench.py --steps 14 --warmup 4 --no-cpu-baseline --no-config2 --parity-mode-steps 0 --no-roofline > gpurun_out/prep_$V.json 2> gpurun_out/prep_$V.err
  python -c "
import json; j=json.load(open('gpurun_out/prep_$V.json')); print('JAF_PREP_AT=$V: %.2f ms/step (median %.2f)' % (j['ms_per_step'], j['median_ms_per_step']))"
done
