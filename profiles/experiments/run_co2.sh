#!/bin/bash
cd $GRAFT_REPO_ROOT
for V in "JAF_AUX_PRIO=1" "JAF_AUX_PRIO=" "JAF_CHAIN_PRIORITY=0" "JAF_AUX_PRIO=1" "JAF_AUX_PRIO=" "JAF_CHAIN_PRIORITY=0"; do
  env $V python bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-config2 --parity-mode-steps 0 --no-roofline > gpurun_out/pr.json 2> gpurun_out/pr.err
  python -c "
import json; j=json.load(open('gpurun_out/pr.json')); print('$V: %.2f ms/step (median %.2f)' % (j['ms_per_step'], j['median_ms_per_step']))"
done
