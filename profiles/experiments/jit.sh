#!/bin/bash
for i in 1 2 3; do
python bench.py --no-cpu-baseline --no-config2 --no-roofline --parity-mode-steps 0 > gpurun_out/jit_$i.json 2>/dev/null
python - <<PY
import json
d = json.load(open("gpurun_out/jit_$i.json"))
print("run $i: mean %.2f median %.2f" % (d["ms_per_step"], d["median_ms_per_step"]))
print("  gpu :", d["step_ms"])
print("  host:", d["host_enqueue_step_ms"])
PY
done
