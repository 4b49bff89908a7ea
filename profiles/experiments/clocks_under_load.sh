#!/bin/bash
# engine / memory clock and power while the default bench runs (rocm-smi sampled twice a second)
cd $GRAFT_REPO_ROOT
python bench.py --steps 120 --warmup 5 --no-cpu-baseline --no-config2 --parity-mode-steps 0 --no-roofline > gpurun_out/clk_bench.json 2> gpurun_out/clk_bench.err &
BP=$!
for i in $(seq 1 50); do
  echo "t=$i $(rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|mclk|Power' | tr '\n' ' ' | sed 's/=\+//g')"
  sleep 0.5
done > gpurun_out/clk_samples.txt 2>&1
wait $BP
tail -c 300 gpurun_out/clk_bench.json
