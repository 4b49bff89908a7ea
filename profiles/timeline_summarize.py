"""Summary of a rocprofv3 --kernel-trace CSV of the multi-stream train step: per hardware queue the busy time, over all queues the
union of busy intervals (time at least one kernel is running), the idle gaps, and the average number of kernels in flight.
Steps are cut at the largest gaps of the adam_kernel launches (last kernels of a step)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    ev.append((s, e, r.get("Queue_Id", "?"), r["Kernel_Name"]))
ev.sort()
t0 = ev[0][0]
# timed region: the last 6 steps = last 6/9 of the kernels roughly; take the last 60 % of the trace by time
T0 = ev[0][0] + int(0.45 * (ev[-1][1] - ev[0][0]))
ev = [x for x in ev if x[0] >= T0]
span = ev[-1][1] - ev[0][0]
# union of busy intervals
busy = 0; cur_s, cur_e = ev[0][0], ev[0][1]; gaps = []
for s, e, q, n in ev[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; gaps.append((s - cur_e, cur_e, n)); cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = sum(e - s for s, e, q, n in ev)
print("window %.2f ms, %d kernels; some kernel running %.2f ms (%.1f %%), idle %.2f ms; sum of kernel durations %.2f ms (%.2f in flight on average while busy)"
      % (span / 1e6, len(ev), busy / 1e6, 100.0 * busy / span, (span - busy) / 1e6, tot / 1e6, tot / max(busy, 1)))
perq = collections.defaultdict(lambda: [0, 0])
for s, e, q, n in ev:
    perq[q][0] += e - s; perq[q][1] += 1
for q, (d, c) in sorted(perq.items(), key=lambda kv: -kv[1][0]):
    print("  queue %-6s busy %.2f ms (%.1f %% of the window), %d kernels" % (q, d / 1e6, 100.0 * d / span, c))
gaps.sort(reverse=True)
print("idle gaps: %d; > 5 us: %d totalling %.2f ms; > 20 us: %d totalling %.2f ms" % (
    len(gaps), sum(1 for g in gaps if g[0] > 5000), sum(g[0] for g in gaps if g[0] > 5000) / 1e6,
    sum(1 for g in gaps if g[0] > 20000), sum(g[0] for g in gaps if g[0] > 20000) / 1e6))
print("largest gaps (us) and the kernel that ended them:")
for g, at, n in gaps[:25]:
    print("  %8.1f  at +%.2f ms  %s" % (g / 1e3, (at - ev[0][0]) / 1e6, n[:90]))
# exclusive time: how long is exactly ONE kernel running (no overlap)?
pts = []
for s, e, q, n in ev:
    pts.append((s, 1)); pts.append((e, -1))
pts.sort()
depth = 0; last = pts[0][0]; hist = collections.Counter()
for t, d in pts:
    hist[depth] += t - last; last = t; depth += d
print("time by number of kernels in flight: " + ", ".join("%d: %.2f ms" % (k, v / 1e6) for k, v in sorted(hist.items()) if v > 0))
