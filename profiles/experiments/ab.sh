#!/bin/bash
# A/B of an environment switch on the default bench: usage ab.sh TAG VAR=VALUE...
TAG=$1; shift
env "$@" python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-config2 --parity-mode-steps 0 > gpurun_out/${TAG}.json 2> gpurun_out/${TAG}.err
python - <<PY
import json
j = json.load(open("gpurun_out/${TAG}.json"))
r = j["roofline"]
print("${TAG}: %.2f ms/step (median %.2f)  all-MFMA %.2f ms" % (j["ms_per_step"], j["median_ms_per_step"], r["all_mfma_kernels"]["ms_per_step"]))
for k, v in sorted(r["by_kernel"].items(), key=lambda kv: -kv[1]["ms"])[:16]:
    print("   %-44s %4d x  %7.3f ms  %7.1f TF" % (k, v["launches"], v["ms"], v["tflops"]))
PY
