"""Reads a rocprofv3 --kernel-trace CSV and prints, for the last submissions, when each kernel started and ended (us, relative):
shows how far the predictor stage of one submission overlaps the Rice stage of the next.  usage: trace_overlap.py <dir>"""
import csv, glob, sys
rows = []
for p in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    with open(p) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
rows = [r for r in rows if r[2].startswith("clx_k_")]
tail = rows[-24:]
t0 = tail[0][0]
for a, b, n in tail:
    print("%-22s start %9.1f  end %9.1f  dur %7.1f us" % (n, (a - t0) / 1e3, (b - t0) / 1e3, (b - a) / 1e3))
