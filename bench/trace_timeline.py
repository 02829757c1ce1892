"""Timeline summary of a rocprofv3 --kernel-trace CSV: for the LAST full training step (from one
launch of the first-layer forward kernel to the next: the optimizer runs per bucket on a side stream since
round 3 and no longer marks the end of a step), wall time, union of kernel-busy time, time with >= 2 kernels in flight, idle
gaps, busy time per queue, and the largest kernels in start order with their overlap partners.
usage: trace_timeline.py <dir with *_kernel_trace.csv>"""
import csv
import glob
import os
import sys

d = sys.argv[1]
f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[0]
rows = list(csv.DictReader(open(f)))
ev = []
for r in rows:
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", r.get("Stream_Id", "?"))))
ev.sort()
first = [i for i, e in enumerate(ev) if "conv1_fprop_kernel" in e[2]]
if len(first) < 2:
    sys.exit("need >= 2 steps (conv1_fprop_kernel launches) in the trace")
lo, hi = first[-2], first[-1]
step = ev[lo:hi]
MIN_US = float(os.environ.get("MIN_US", "300"))
t0, t1 = step[0][0], max(e[1] for e in step)
print("# one training step from the kernel trace (%d kernels), times in ms" % len(step))
print("wall %.2f" % ((t1 - t0) / 1e6))
pts = []
for s, e, _, _ in step:
    pts.append((s, 1)); pts.append((e, -1))
pts.sort()
busy = over = 0
depth = 0
last = pts[0][0]
for t, dlt in pts:
    if depth >= 1:
        busy += t - last
    if depth >= 2:
        over += t - last
    depth += dlt
    last = t
print("union busy %.2f, idle %.2f, >= 2 kernels in flight %.2f, sum of kernel times %.2f" % (
    busy / 1e6, (t1 - t0 - busy) / 1e6, over / 1e6, sum(e - s for s, e, _, _ in step) / 1e6))
q = {}
for s, e, _, qq in step:
    q[qq] = q.get(qq, 0) + e - s
print("busy per queue: " + ", ".join("%s: %.2f" % (k, v / 1e6) for k, v in sorted(q.items())))
# idle gaps > 20 us
gaps = []
cur_end = step[0][1]
for s, e, n, _ in step[1:]:
    if s > cur_end + 20000:
        gaps.append((s - cur_end, cur_end - t0, n))
    cur_end = max(cur_end, e)
print("idle gaps > 20 us: %d, total %.2f" % (len(gaps), sum(g[0] for g in gaps) / 1e6))
for g in sorted(gaps, reverse=True)[:8]:
    print("  gap %.3f at +%.2f before %s" % (g[0] / 1e6, g[1] / 1e6, g[2][:60]))
print("\n| start | dur | queue | kernel |\n|---|---|---|---|")
for s, e, n, qq in step:
    if e - s > MIN_US * 1e3:
        print("| %.2f | %.2f | %s | `%s` |" % ((s - t0) / 1e6, (e - s) / 1e6, qq, n[:70]))
