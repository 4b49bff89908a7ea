#!/bin/bash
# HBM traffic of the bf16 step: FETCH_SIZE and WRITE_SIZE in separate passes (MI355X_MICROARCH.md, HBM section)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 280 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_traffic_$c -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-config2 --parity-mode-steps 0 --no-frame-parity --serial-streams $JAF_PROFILE_ARGS > $R/gpurun_out/pmc_traffic_$c.log 2>&1; echo "$c rc=$?"
done
