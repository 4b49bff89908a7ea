#!/bin/bash
# Round-3 artefacts, taken on one MI355X box in one gpurun call:  bash profiles/round3.sh <tag>
#   <tag>_kernel_stats_bf16.csv, <tag>_bench_under_rocprof.json   rocprofv3 --kernel-trace --stats, everything on one stream
#   <tag>_pmc_hbm_traffic.txt, pmc_traffic.json                   rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes)
#   <tag>_bench_bf16.json                                         the default `python bench.py` line
#   <tag>_bench_graph.json                                        `python bench.py --graph` (hipGraph replay)
#   <tag>_host_enqueue.txt                                        host enqueue time vs step time, eager and graph
#   <tag>_pytest_durations.txt                                    pytest -m gpu --durations=40
TAG=${1:-round3_a}
R=$GRAFT_REPO_ROOT
cd $R
bash profiles/kernel_stats.sh $TAG > gpurun_out/${TAG}_kernel_stats.log 2>&1
cp gpurun_out/${TAG}_kernel_stats.csv gpurun_out/${TAG}_kernel_stats_bf16.csv
bash profiles/pmc_traffic.sh > gpurun_out/${TAG}_pmc.log 2>&1
python3 profiles/pmc_summarize.py $TAG >> gpurun_out/${TAG}_pmc.log 2>&1
cp profiles/${TAG}_pmc_hbm_traffic.txt profiles/pmc_traffic.json gpurun_out/ 2>/dev/null
rm -rf gpurun_out/pmc_traffic_FETCH_SIZE gpurun_out/pmc_traffic_WRITE_SIZE
cd $R
python3 bench.py > gpurun_out/${TAG}_bench_bf16.json 2> gpurun_out/${TAG}_bench.err
python3 bench.py --graph --no-cpu-baseline --no-config2 --parity-mode-steps 0 > gpurun_out/${TAG}_bench_graph.json 2> gpurun_out/${TAG}_bench_graph.err
python3 profiles/host_enqueue.py 2>&1 | head -12 > gpurun_out/${TAG}_host_enqueue.txt
python3 -m pytest tests -m gpu -q --durations=40 -p no:cacheprovider 2>&1 | tail -60 > gpurun_out/${TAG}_pytest_durations.txt
tail -3 gpurun_out/${TAG}_pytest_durations.txt
