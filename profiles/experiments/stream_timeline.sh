#!/bin/bash
# usage (on the GPU box): bash profiles/experiments/stream_timeline.sh <tag>  -> gpurun_out/<tag>_stream_timeline.txt
TAG=${1:-tl}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl_$TAG -o p -- python3 $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-roofline --no-config2 --parity-mode-steps 0 > $R/gpurun_out/${TAG}_bench_under_trace.json 2> $R/gpurun_out/${TAG}_trace.log
echo "rc=$?"
F=$(find $R/gpurun_out/tl_$TAG -name "*kernel_trace.csv" | head -1)
head -1 $F
python3 $R/profiles/experiments/stream_timeline.py $F 3 8 $R/gpurun_out/${TAG}_step_sequence.txt > $R/gpurun_out/${TAG}_stream_timeline.txt 2>&1
rm -rf $R/gpurun_out/tl_$TAG
cat $R/gpurun_out/${TAG}_stream_timeline.txt
