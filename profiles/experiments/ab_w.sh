#!/bin/bash
# A/B on the default bench, printing the weight-gradient rows: ab_w.sh TAG VAR=VALUE...
TAG=$1; shift
env "$@" python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-config2 --parity-mode-steps 0 > gpurun_out/${TAG}.json 2> gpurun_out/${TAG}.err
python - <<PY
import json
j = json.load(open("gpurun_out/${TAG}.json"))
r = j["roofline"]
w = {k: v for k, v in r["by_kernel"].items() if "wgrad" in k}
print("${TAG}: %.2f ms/step (median %.2f)  wgrad total %.2f ms" % (j["ms_per_step"], j["median_ms_per_step"], sum(v["ms"] for v in w.values())))
for k, v in sorted(w.items(), key=lambda kv: -kv[1]["ms"]):
    print("   %-52s %4d x  %7.3f ms  %7.1f TF" % (k, v["launches"], v["ms"], v["tflops"]))
PY
