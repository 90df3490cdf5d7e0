"""Copy a round's rocprofv3 summaries from gpurun_out/prof_<ROUND>_<name>/ (tools/profile_round.sh) into profiles/ and renew the round's entries of
profiles/pmc_traffic.json (tools/update_traffic.py).  usage: python tools/collect_profiles.py [ROUND=r06] [what the source text says about the build]"""
import csv, glob, json, os, shutil, subprocess, sys
from collections import defaultdict
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
rnd = sys.argv[1] if len(sys.argv) > 1 else "r06"
build = sys.argv[2] if len(sys.argv) > 2 else "round 6 final build"
keys = {"config3": "config3_lanes_frames_10000", "config2": "config2_lanes_frames_10000", "config4": "config4_lanes_frames_10000",
        "config5": "config5_lanes_frames_10001", "config5share": "config5_lanes_frames_125003"}
for name, key in keys.items():
    out = os.path.join(ROOT, "gpurun_out", "prof_%s_%s" % (rnd, name))
    if not os.path.exists(os.path.join(out, "summary.txt")):
        continue
    P = lambda s: os.path.join(ROOT, "profiles", "%s_%s_%s" % (rnd, name, s))
    shutil.copy(os.path.join(out, "summary.txt"), P("summary.txt"))
    for src, dst in (("trace/trace_kernel_stats.csv", "kernel_stats.csv"), ("merged12/t_kernel_stats.csv", "merged12_kernel_stats.csv"),
                     ("pipe/t_kernel_stats.csv", "pipelined_kernel_stats.csv"), ("pipelined_trace.txt", "pipelined_trace.txt")):
        f = glob.glob(os.path.join(out, "**", os.path.basename(src)), recursive=True) if "/" in src else [os.path.join(out, src)]
        f = [x for x in f if os.path.dirname(src) in x and os.path.exists(x)]
        if f:
            shutil.copy(f[0], P(dst))
    src_text = "profiles/%s_%s_summary.txt (rocprofv3 --pmc passes of ROUND=%s tools/profile_round.sh %s, %s; steps one at a time)" % (rnd, name, rnd, name, build)
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "update_traffic.py"), out, key, src_text])
    if name == "config3":      # HBM traffic of ONE merged launch of twelve runs (per run)
        vals = defaultdict(lambda: defaultdict(list))
        for cnt in ("FETCH_SIZE", "WRITE_SIZE"):
            for f in glob.glob(os.path.join(out, "merged12_" + cnt, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if row.get("Kernel_Name", "").startswith("clx_k_"):
                        vals[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
        if vals:
            line = json.loads(open(os.path.join(out, "bench_line.json")).read())
            alg = line["roofline"]["algorithmic_bytes_per_launch"]
            mean = lambda v: sum(v) / len(v) if v else 0.0
            tot = 0.0
            with open(P("merged12_traffic.txt"), "w") as fh:
                fh.write("# HBM traffic of ONE merged launch of twelve runs of config 3 (the launch shape of the timed steps): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in\n"
                         "# separate passes of tools/merge_probe.py 12 3 (tools/profile_round.sh config3; %s), bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024, per run\n" % build)
                for k in sorted(vals):
                    b = (2 * mean(vals[k]["FETCH_SIZE"]) + mean(vals[k]["WRITE_SIZE"])) * 1024 / 12.0
                    tot += b
                    fh.write("%-22s %7.1f MB per run   (fetch %.1f, write %.1f)\n" % (k, b / 1e6, 2 * mean(vals[k]["FETCH_SIZE"]) * 1024 / 12e6, mean(vals[k]["WRITE_SIZE"]) * 1024 / 12e6))
                fh.write("total %.1f MB per run = %.3f x the algorithmic %.1f MB (steps one at a time: profiles/%s_config3_summary.txt)\n" % (tot / 1e6, tot / alg, alg / 1e6, rnd))
            print(open(P("merged12_traffic.txt")).read())
