"""Summarise rocprofv3 CSV output (kernel-trace stats + PMC passes) into a small text table."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(pattern):
    return sorted(glob.glob(os.path.join(out, "**", pattern), recursive=True))


print("# rocprofv3 summary for", os.path.basename(out))
for f in find("*kernel_stats.csv"):
    print("\n## kernel stats (%s)" % os.path.relpath(f, out))
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row.get("Name", "")
            if name.startswith("clx_") or "clx_" in name:
                print("%-22s calls=%s avg_ns=%s min_ns=%s max_ns=%s total_ns=%s pct=%s" % (
                    name[:22], row.get("Calls"), row.get("AverageNs"), row.get("MinNs"), row.get("MaxNs"),
                    row.get("TotalDurationNs"), row.get("Percentage")))
for f in find("*kernel_trace.csv"):
    if "/trace/" not in f:
        continue
    durs = defaultdict(list)
    meta = {}
    with open(f) as fh:
        for row in csv.DictReader(fh):
            n = row.get("Kernel_Name", "")
            if "clx_" not in n:
                continue
            durs[n].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
            meta[n] = (row.get("VGPR_Count"), row.get("Accum_VGPR_Count"), row.get("SGPR_Count"), row.get("LDS_Block_Size"),
                       row.get("Grid_Size"), row.get("Workgroup_Size"))
    print("\n## kernel trace durations (ns)")
    for n, d in durs.items():
        d2 = sorted(d)
        print("%-22s n=%d mean=%.0f median=%d min=%d max=%d  vgpr/agpr/sgpr/lds/grid/wg=%s" % (
            n[:22], len(d), sum(d) / len(d), d2[len(d2) // 2], d2[0], d2[-1], meta[n]))
for f in find("*counter_collection.csv"):
    acc = defaultdict(lambda: defaultdict(list))
    with open(f) as fh:
        for row in csv.DictReader(fh):
            n = row.get("Kernel_Name", "")
            if "clx_" not in n:
                continue
            acc[n][row["Counter_Name"]].append(float(row["Counter_Value"]))
    print("\n## counters (%s) -- mean per dispatch" % os.path.relpath(f, out))
    for n, cs in acc.items():
        for c, v in sorted(cs.items()):
            print("%-22s %-24s %.4g  (n=%d)" % (n[:22], c, sum(v) / len(v), len(v)))
