#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_modules.py -q -x -k "batchnorm or split_batchnorm or discriminator or bn" -p no:cacheprovider 2>&1 | tail -2
for V in 1 0 1 0; do
  JAF_SPLIT_BN_ONE_LAUNCH=$V python bench.py --steps 14 --warmup 4 --no-cpu-baseline --no-config2 --parity-mode-steps 0 --no-roofline > gpurun_out/bn_$V.json 2> gpurun_out/bn_$V.err
  python -c "
import json; j=json.load(open('gpurun_out/bn_$V.json')); print('JAF_SPLIT_BN_ONE_LAUNCH=$V: %.2f ms/step (median %.2f) host enqueue %.1f' % (j['ms_per_step'], j['median_ms_per_step'], j['config'].get('host_enqueue_ms', 0)))"
done
