#!/bin/bash
# rocprofv3 --kernel-trace --stats of the bf16 step with everything on one stream (per-kernel durations free of overlap)
# usage (on the GPU box): bash profiles/kernel_stats.sh <tag>   -> gpurun_out/<tag>_kernel_stats.csv, <tag>_bench_under_rocprof.json
TAG=${1:-stats}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o p -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-roofline --no-config2 --parity-mode-steps 0 --no-frame-parity --serial-streams $JAF_PROFILE_ARGS > $R/gpurun_out/${TAG}_bench_under_rocprof.json 2> $R/gpurun_out/${TAG}_rocprof.log
echo "rc=$?"
cp $(find $R/gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${TAG}_kernel_stats.csv
rm -rf $R/gpurun_out/prof_$TAG
head -45 $R/gpurun_out/${TAG}_kernel_stats.csv | cut -c1-150
