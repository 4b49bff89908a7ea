#!/bin/bash
# rocprofv3 --kernel-trace of the multi-stream bf16 step (timestamps per kernel and queue) -> gpurun_out/<tag>_timeline.txt
# (profiles/timeline_summarize.py: per-queue busy time, union of busy intervals, idle gaps of the whole GPU per step)
# usage (on the GPU box): bash profiles/timeline.sh <tag>
TAG=${1:-tl}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_$TAG -o p -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --no-config2 --parity-mode-steps 0 --no-frame-parity > $R/gpurun_out/${TAG}_bench_under_trace.json 2> $R/gpurun_out/${TAG}_trace.log
echo "rc=$?"
F=$(find $R/gpurun_out/prof_$TAG -name "*kernel_trace.csv" | head -1)
python3 $R/profiles/timeline_summarize.py $F > $R/gpurun_out/${TAG}_timeline.txt
rm -rf $R/gpurun_out/prof_$TAG
cat $R/gpurun_out/${TAG}_timeline.txt
