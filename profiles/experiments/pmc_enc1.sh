#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python3 $R/scratch/mb_enc1.py 3 2>&1 | grep -v amdgpu | tail -8
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  timeout 180 rocprofv3 --pmc $line --output-format csv -d $R/gpurun_out/pmc_enc1_$i -o p -- python3 $R/scratch/mb_enc1.py 2 > $R/gpurun_out/pmc_enc1_$i.log 2>&1; echo "pass $i rc=$?"
done <<'LIST'
FETCH_SIZE
WRITE_SIZE
TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU
SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM
LIST
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("$R/gpurun_out/pmc_enc1_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:72]
        if "wgrad" not in k and "conv_dma" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
with open("$R/gpurun_out/pmc_enc1.txt", "w") as out:
    for k, v in agg.items():
        out.write(k + "\n")
        for c, x in sorted(v.items()): out.write("    %-36s %16.0f  (n=%d)\n" % (c, x / cnt[(k, c)], cnt[(k, c)]))
PY
rm -rf $R/gpurun_out/pmc_enc1_[0-9]
cat $R/gpurun_out/pmc_enc1.txt
