#!/bin/bash
cd /root/repo
out=gpurun_out/ab_rank1b.txt; : > $out
run() { name=$1; shift; r=$(env "$@" python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-config2 --no-roofline --parity-mode-steps 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config'].get('host_enqueue_ms'))"); echo "$name $r" | tee -a $out; }
for i in 1 2; do
  run g1_q3 JAF_BENCH_ONE_RANK_GROUP=1 GPU_MAX_HW_QUEUES=3
  run g1_noprio JAF_BENCH_ONE_RANK_GROUP=1 JAF_CHAIN_PRIORITY=0
  run g1_q3_noprio JAF_BENCH_ONE_RANK_GROUP=1 GPU_MAX_HW_QUEUES=3 JAF_CHAIN_PRIORITY=0
  run g1_hiprio JAF_BENCH_ONE_RANK_GROUP=1 TORCH_NCCL_HIGH_PRIORITY=1
  run g1_q2 JAF_BENCH_ONE_RANK_GROUP=1 GPU_MAX_HW_QUEUES=2
done
