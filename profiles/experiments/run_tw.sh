#!/bin/bash
cd $GRAFT_REPO_ROOT
for V in "JAF_X=0" "JAF_PLAN_TW=32" "JAF_PLAN_TW=64" "JAF_X=0" "JAF_PLAN_TW=32" "JAF_PLAN_TW=64"; do
  env $V python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-config2 --parity-mode-steps 0 > gpurun_out/tw_$V.json 2> gpurun_out/tw_$V.err
  python - <<PY
import json
j = json.load(open("gpurun_out/tw_$V.json"))
r = j["roofline"]["by_kernel"]
w = {k: v for k, v in r.items() if "conv_dma_kernel" in k}
print("$V: %.2f ms/step (median %.2f)  conv_dma total %.2f" % (j["ms_per_step"], j["median_ms_per_step"], sum(v["ms"] for v in w.values())))
for k, v in sorted(w.items(), key=lambda kv: -kv[1]["ms"])[:9]:
    print("   %-52s %4d x  %7.3f ms" % (k, v["launches"], v["ms"]))
PY
done
