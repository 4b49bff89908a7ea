#!/bin/bash
# usage: pmc_w.sh <tag> <layer of mb_wgrad.py> ; SQ counter passes for the fwd / dgrad / wgrad kernels of one layer
cd /tmp && export TMPDIR=/tmp
TAG=$1; L=$2
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $line --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_$i -o p -- python3 $GRAFT_REPO_ROOT/scratch/mb_wgrad.py bf16 $L 3 > $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_$i.log 2>&1; echo "pass $i rc=$?"
done <<'LIST'
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU
SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM
GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum
LIST
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("$GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:64]
        if "conv" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
with open("$GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}.txt", "w") as out:
    for k, v in agg.items():
        out.write(k + "\n")
        for c, x in sorted(v.items()): out.write("    %-36s %16.0f  (n=%d)\n" % (c, x / cnt[(k, c)], cnt[(k, c)]))
PY
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_[0-9]
cat $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}.txt
