#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_kernels.py -q -x -k "conv2d or lstm or kernel_names" -p no:cacheprovider 2>&1 | tail -2
python -m pytest tests/test_gpu_step.py tests/test_gpu_step_parity.py -q -x -k "bf16x3" -p no:cacheprovider 2>&1 | tail -2
for V in 1 0 1 0; do
  JAF_WGRAD_FAST_SPLIT=$V python bench.py --precision bf16x3 --steps 10 --warmup 4 --no-cpu-baseline --no-config2 --parity-mode-steps 0 > gpurun_out/fs_$V.json 2> gpurun_out/fs_$V.err
  python - <<PY
import json
j = json.load(open("gpurun_out/fs_$V.json"))
r = j["roofline"]["by_kernel"]
w = {k: v for k, v in r.items() if "wgrad" in k}
print("FAST_SPLIT=$V: %.2f ms/step (median %.2f)  wgrad total %.2f" % (j["ms_per_step"], j["median_ms_per_step"], sum(v["ms"] for v in w.values())))
for k, v in sorted(w.items(), key=lambda kv: -kv[1]["ms"])[:6]:
    print("   %-56s %4d x  %7.3f ms  %7.1f TF" % (k, v["launches"], v["ms"], v["tflops"]))
PY
done
