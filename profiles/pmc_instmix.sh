#!/bin/bash
# Instruction mix of every kernel of the bf16 step (one rocprofv3 --pmc pass per counter group, no tracing): bash profiles/pmc_instmix.sh <tag>
# -> gpurun_out/<tag>_pmc_instmix.txt: per kernel (sum over 3 steps / launches): VALU, MFMA, LDS, VMEM instructions per wave, share of the
# wave time spent issuing / waiting for operands / in s_waitcnt + barriers (SQ_*_CYCLES count quad-cycles).
TAG=${1:-instmix}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  timeout 280 rocprofv3 --pmc $line --output-format csv -d $R/gpurun_out/pmc_mix_$i -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-config2 --parity-mode-steps 0 --no-frame-parity --serial-streams $JAF_PROFILE_ARGS > $R/gpurun_out/pmc_mix_$i.log 2>&1; echo "pass $i rc=$?"
done <<'LIST'
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU
SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_INSTS_VMEM_WR
LIST
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob("$R/gpurun_out/pmc_mix_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        k = (k[5:] if k.startswith("void ") else k).split("(")[0]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] in ("SQ_WAVES",): n[k] += 1
rows = []
for k, v in agg.items():
    w = max(v.get("SQ_WAVES", 0), 1.0); wc = max(v.get("SQ_WAVE_CYCLES", 0), 1.0)
    rows.append((wc, k, n[k], v.get("SQ_INSTS_VALU", 0) / w, v.get("SQ_INSTS_MFMA", 0) / w, v.get("SQ_INSTS_LDS", 0) / w,
                 (v.get("SQ_INSTS_VMEM_RD", 0) + v.get("SQ_INSTS_VMEM_WR", 0)) / w, v.get("SQ_INSTS_SALU", 0) / w,
                 v.get("SQ_ACTIVE_INST_ANY", 0) / wc, v.get("SQ_WAIT_INST_ANY", 0) / wc, v.get("SQ_WAIT_ANY", 0) / wc,
                 v.get("SQ_LDS_BANK_CONFLICT", 0) / max(v.get("SQ_ACTIVE_INST_LDS", 0), 1.0), 4.0 * wc / w))
rows.sort(reverse=True)
with open("$R/gpurun_out/${TAG}_pmc_instmix.txt", "w") as out:
    out.write("%-64s %6s %8s %7s %7s %7s %7s %6s %6s %6s %6s %9s\n" % ("kernel (by total wave time)", "calls", "VALU/wv", "MFMA/wv", "LDS/wv", "VMEM/wv", "SALU/wv", "issue", "w.inst", "w.any", "confl", "cyc/wave"))
    for r in rows[:60]:
        out.write("%-64s %6d %8.0f %7.0f %7.0f %7.0f %7.0f %6.2f %6.2f %6.2f %6.2f %9.0f\n" % ((r[1][:64],) + r[2:]))
PY
rm -rf $R/gpurun_out/pmc_mix_[0-9]
head -45 $R/gpurun_out/${TAG}_pmc_instmix.txt
