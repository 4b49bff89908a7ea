#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_modules.py -q -x -k "lazy or crn or CRN or packed" -p no:cacheprovider 2>&1 | tail -3
for V in "JAF_PACK_UP=1" "JAF_PACK_UP=0" "JAF_PACK_UP=1" "JAF_PACK_UP=0"; do
  env $V python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-config2 --parity-mode-steps 0 > gpurun_out/up_$V.json 2> gpurun_out/up_$V.err
  python - <<PY
import json
j = json.load(open("gpurun_out/up_$V.json"))
r = j["roofline"]["by_kernel"]
w = {k: v for k, v in r.items() if "pack" in k}
print("$V: %.2f ms/step (median %.2f)" % (j["ms_per_step"], j["median_ms_per_step"]))
for k, v in sorted(w.items(), key=lambda kv: -kv[1]["ms"])[:6]:
    print("   %-52s %4d x  %7.3f ms  %s" % (k, v["launches"], v["ms"], v.get("gbps", "")))
PY
done
