#!/bin/bash
# Round-5 artefacts, one gpurun call:  bash profiles/round5.sh <tag> [256|512] [quick]
#   <tag>_kernel_stats_bf16.csv, <tag>_bench_under_rocprof.json   rocprofv3 --kernel-trace --stats, everything on one stream
#   <tag>_pmc_hbm_traffic.txt, pmc_traffic[_512].json             rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes)
#   <tag>_bench_bf16.json                                         the default `python bench.py [--size 512]` line
# 256 only, unless `quick`:  <tag>_host_enqueue.txt, <tag>_pytest_durations.txt
TAG=${1:-round5_a}
SIZE=${2:-256}
QUICK=${3:-}
export JAF_PROFILE_COMMIT=${4:-unknown}      # HEAD of the build container at gpurun time (the box has no .git)
R=$GRAFT_REPO_ROOT
cd $R
if [ "$SIZE" != "256" ]; then export JAF_PROFILE_ARGS="--size $SIZE"; JNAME=pmc_traffic_$SIZE.json; else export JAF_PROFILE_ARGS=""; JNAME=pmc_traffic.json; fi
bash profiles/kernel_stats.sh $TAG > gpurun_out/${TAG}_kernel_stats.log 2>&1
cp gpurun_out/${TAG}_kernel_stats.csv gpurun_out/${TAG}_kernel_stats_bf16.csv
# the matrix-core kernel with the largest total time of that trace = the row bench.py reports in `roofline`
DOM=$(python3 - <<PY
import csv
rows = [r for r in csv.DictReader(open("gpurun_out/${TAG}_kernel_stats.csv")) if "conv_dma_kernel" in r["Name"] or "conv_wgrad_dma_kernel" in r["Name"]]
r = max(rows, key=lambda r: float(r["TotalDurationNs"]))
n = r["Name"]
print((n[5:] if n.startswith("void ") else n).split("(")[0])
PY
)
echo "dominant matrix-core kernel: $DOM"
bash profiles/pmc_traffic.sh > gpurun_out/${TAG}_pmc.log 2>&1
python3 profiles/pmc_summarize.py $TAG "$DOM" $JNAME "$JAF_PROFILE_ARGS" >> gpurun_out/${TAG}_pmc.log 2>&1
cp profiles/${TAG}_pmc_hbm_traffic.txt profiles/$JNAME gpurun_out/ 2>/dev/null
rm -rf gpurun_out/pmc_traffic_FETCH_SIZE gpurun_out/pmc_traffic_WRITE_SIZE
python3 bench.py $JAF_PROFILE_ARGS > gpurun_out/${TAG}_bench_bf16.json 2> gpurun_out/${TAG}_bench.err
tail -c 600 gpurun_out/${TAG}_bench_bf16.json
if [ "$SIZE" == "256" ] && [ -z "$QUICK" ]; then
  python3 profiles/host_enqueue.py 2>&1 | head -12 > gpurun_out/${TAG}_host_enqueue.txt
  python3 -m pytest tests -m gpu -q --durations=40 -p no:cacheprovider 2>&1 | tail -60 > gpurun_out/${TAG}_pytest_durations.txt
  tail -3 gpurun_out/${TAG}_pytest_durations.txt
fi
