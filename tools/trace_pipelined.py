"""Kernel trace of the TIMED mode of bench.py (steps submitted with several in flight): reads a rocprofv3 --kernel-trace CSV and
prints, for the steady part of the run, every clx_k_* kernel's start / end / duration, the queue and stream it ran on, how many
scan / decode kernels were running at the same time, and the ratio of a kernel's duration here to its duration alone.
usage: trace_pipelined.py <dir> [kernel_ms_alone.json]"""
import csv, glob, json, sys
rows = []
for p in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    with open(p) as f:
        for r in csv.DictReader(f):
            if r["Kernel_Name"].startswith("clx_k_"):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Stream_Id", "?"),
                             r.get("VGPR_Count", "?"), r.get("LDS_Block_Size", "?")))
rows.sort()
if not rows:
    sys.exit("no clx_k_* kernels in the trace")
# the timed loop = the longest stretch without a gap > 1 ms
groups, cur = [], [rows[0]]
for r in rows[1:]:
    if r[0] - max(x[1] for x in cur) > 1_000_000:
        groups.append(cur); cur = []
    cur.append(r)
groups.append(cur)
# the timed steps are the LAST thing bench.py runs (behind the per-kernel profiling phases and the clearing of the output buffers):
# the last stretch that holds decode kernels on more than one queue
multi = [c for c in groups if len(set(x[3] for x in c if x[2] in ("clx_k_lean", "clx_k_lean24", "clx_k_lanes", "clx_k_residual"))) > 1]
g = multi[-1] if multi else groups[-1]
t0, t1 = g[0][0], max(x[1] for x in g)
names = sorted(set(r[2] for r in g))
print("# timed stretch: %d kernels, %.3f ms wall" % (len(g), (t1 - t0) / 1e6))
print("# queues used: %s   streams: %s" % (sorted(set(r[3] for r in g)), len(set(r[4] for r in g))))
for n in names:
    d = sorted((r[1] - r[0]) / 1e3 for r in g if r[2] == n)
    meta = next(r for r in g if r[2] == n)
    print("%-18s n=%3d  dur us: min %8.1f  median %8.1f  max %8.1f   vgpr %s lds %s" % (n, len(d), d[0], d[len(d) // 2], d[-1], meta[5], meta[6]))
# concurrency: at each kernel start, how many kernels of each name are running
big = [n for n in names if n in ("clx_k_scan", "clx_k_lean", "clx_k_lanes", "clx_k_residual", "clx_k_predict16")]
for n in big:
    ev = []
    for r in g:
        if r[2] == n:
            ev.append((r[0], 1)); ev.append((r[1], -1))
    ev.sort()
    c = 0; area = 0.0; last = ev[0][0]; mx = 0
    for t, dlt in ev:
        area += c * (t - last); last = t; c += dlt; mx = max(mx, c)
    print("%-18s running at once: time-average %.2f over the stretch, max %d" % (n, area / (t1 - t0), mx))
print("# steady middle of the stretch (us from its start):")
mid = [r for r in g if r[2] in big][len(g) // 3: len(g) // 3 + 36]
for a, b, n, q, s, _, _ in mid:
    print("%-16s q%-3s s%-3s start %9.1f  end %9.1f  dur %7.1f" % (n, q, s, (a - t0) / 1e3, (b - t0) / 1e3, (b - a) / 1e3))
if len(sys.argv) > 2 and sys.argv[2] == "all":
    # every kernel of the LAST timed region (behind the stretch's last gap of more than 0.2 ms with nothing running)
    cuts = [0]
    hi = g[0][1]
    for i, r in enumerate(g[1:], 1):
        if r[0] - hi > 200_000: cuts.append(i)
        hi = max(hi, r[1])
    last = g[cuts[-1]:]
    z = last[0][0]
    print("# the last region, every kernel (us from its start):")
    for a, b, n, q, s, _, _ in last:
        print("%-20s q%-3s start %9.1f  end %9.1f  dur %7.1f" % (n, q, (a - z) / 1e3, (b - z) / 1e3, (b - a) / 1e3))
