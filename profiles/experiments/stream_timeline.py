"""Per-queue timeline of the multi-stream step from a rocprofv3 kernel trace (run ON the GPU box by stream_timeline.sh):
busy time per HIP stream and step, the idle gaps of the main stream, and which kernels sit on it.
usage: python3 stream_timeline.py <kernel_trace.csv> <warmup> <steps>"""
import csv, sys, collections
path, W, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
rows = list(csv.DictReader(open(path)))
skey = "Stream_Id" if "Stream_Id" in rows[0] and len({r["Stream_Id"] for r in rows}) > 1 else "Queue_Id"
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r[skey], r["Kernel_Name"]) for r in rows]
ev.sort()
adam = [e for e in ev if e[3].startswith("adam_kernel")]
per_step = len(adam) // (W + K)
t0 = adam[per_step * W - 1][1]
t1 = adam[per_step * (W + K) - 1][1]
win = [e for e in ev if e[0] >= t0 and e[1] <= t1]
print("key %s, %d Adam launches per step, window %.2f ms = %.2f ms/step, %d kernels/step" % (
    skey, per_step, (t1 - t0) / 1e6, (t1 - t0) / 1e6 / K, len(win) // K))
byq = collections.defaultdict(list)
for e in win:
    byq[e[2]].append(e)
order = sorted(byq, key=lambda q: -sum(e[1] - e[0] for e in byq[q]))
for q in order:
    es = byq[q]
    busy = sum(e[1] - e[0] for e in es) / 1e6 / K
    print("stream %-6s busy %7.2f ms/step  %5d kernels/step" % (q, busy, len(es) // K))
main = byq[order[0]]
gaps = []
for a, b in zip(main, main[1:]):
    g = b[0] - a[1]
    if g > 0:
        gaps.append((g, a[3], b[3]))
tot = sum(g for g, _, _ in gaps) / 1e6 / K
print("main stream: idle %.2f ms/step in %d gaps/step" % (tot, len(gaps) // K))
for lo, hi in ((0, 2e3), (2e3, 1e4), (1e4, 5e4), (5e4, 2e5), (2e5, 1e12)):
    sel = [g for g, _, _ in gaps if lo <= g < hi]
    print("  gaps %6.0f-%-8.0f us: %5d/step  %6.2f ms/step" % (lo / 1e3, hi / 1e3, len(sel) // K, sum(sel) / 1e6 / K))
agg = collections.defaultdict(lambda: [0, 0])
for g, a, b in gaps:
    if g >= 1e4:
        k = (a.split("(")[0][:60], b.split("(")[0][:60])
        agg[k][0] += g; agg[k][1] += 1
print("largest gap sites (>= 10 us), per step:")
for k, (g, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:25]:
    print("  %7.3f ms %4.1fx  after %-60s before %s" % (g / 1e6 / K, n / K, k[0], k[1]))
kt = collections.defaultdict(lambda: [0, 0])
for e in main:
    kt[e[3].split("(")[0][:90]][0] += e[1] - e[0]; kt[e[3].split("(")[0][:90]][1] += 1
print("main stream kernels, per step:")
for k, (t, n) in sorted(kt.items(), key=lambda kv: -kv[1][0])[:45]:
    print("  %7.3f ms %5.1fx  %s" % (t / 1e6 / K, n / K, k))
for q in order[1:]:
    kt = collections.defaultdict(lambda: [0, 0])
    for e in byq[q]:
        kt[e[3].split("(")[0][:90]][0] += e[1] - e[0]; kt[e[3].split("(")[0][:90]][1] += 1
    print("stream %s kernels, per step:" % q)
    for k, (t, n) in sorted(kt.items(), key=lambda kv: -kv[1][0])[:8]:
        print("  %7.3f ms %5.1fx  %s" % (t / 1e6 / K, n / K, k))
# one step's launch sequence (the last one of the window): start (us from the step's first launch), duration, stream, grid, kernel
if len(sys.argv) > 4:
    ts = adam[per_step * (W + K - 1) - 1][1]
    rws = [r for r in rows if int(r["Start_Timestamp"]) >= ts and int(r["End_Timestamp"]) <= t1]
    rws.sort(key=lambda r: int(r["Start_Timestamp"]))
    with open(sys.argv[4], "w") as f:
        for r in rws:
            wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
            nwg = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // max(wg, 1)
            f.write("%9.1f %7.1f s%s wg%-6d %s\n" % ((int(r["Start_Timestamp"]) - ts) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                                                  r[skey], nwg, r["Kernel_Name"].split("(")[0][:100]))
