#!/bin/bash
cd $GRAFT_REPO_ROOT
for V in "JAF_X=1" "JAF_BCE_PAIR=0 JAF_LINEAR_FUSED_BWD=0" "JAF_X=1" "JAF_BCE_PAIR=0 JAF_LINEAR_FUSED_BWD=0" "JAF_X=1" "JAF_BCE_PAIR=0 JAF_LINEAR_FUSED_BWD=0"; do
  env $V python bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-config2 --parity-mode-steps 0 --no-roofline > gpurun_out/bp.json 2> gpurun_out/bp.err
  python -c "
import json; j=json.load(open('gpurun_out/bp.json')); print('$V: %.2f ms/step (median %.2f) host enqueue %.1f' % (j['ms_per_step'], j['median_ms_per_step'], j['config'].get('host_enqueue_ms', 0)))"
done
