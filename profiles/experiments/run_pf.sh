#!/bin/bash
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
JAF_CONV_FAST3=1 python -m pytest tests/test_gpu_kernels.py -q -x -k "conv2d" -p no:cacheprovider 2>&1 | tail -2
for L in crn256 crn512_64 crn32; do for V in "JAF_CONV_FAST3=0" "JAF_CONV_FAST3=1" "JAFPRO_HIP_LIB=scratch/x/lib_head.so"; do echo "== $L $V: $(env $V python profiles/experiments/mb_conv.py bf16 $L 10 2>&1 | grep -v amdgpu | tr '\n' ';')"; done; done
for V in "JAF_CONV_FAST3=1" "JAFPRO_HIP_LIB=scratch/x/lib_head.so" "JAF_CONV_FAST3=1" "JAFPRO_HIP_LIB=scratch/x/lib_head.so"; do
  env $V python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-config2 --parity-mode-steps 0 > gpurun_out/f3.json 2> gpurun_out/f3.err
  python - <<PY
import json
j = json.load(open("gpurun_out/f3.json"))
r = j["roofline"]["by_kernel"]
w = {k: v for k, v in r.items() if "conv_dma_kernel<4, 4" in k}
print("$V: %.2f ms/step (median %.2f)  <4,4,*> total %.2f  " % (j["ms_per_step"], j["median_ms_per_step"], sum(v["ms"] for v in w.values())), {k[15:]: round(v["ms"], 3) for k, v in w.items()})
PY
done
