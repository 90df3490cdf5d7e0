"""profiles/pmc_traffic.json entries from a round-3 profile directory (tools/profile_round.sh): HBM bytes per launch per kernel
((2 x FETCH_SIZE + WRITE_SIZE) x 1024, separate --pmc passes) and the instruction counts of the same kernels (SQ_INSTS_VALU / SALU /
LDS, wave-instructions per launch).  usage: update_traffic.py <gpurun_out/prof_r03_NAME> <key> <source text>"""
import csv, glob, json, os, sys
from collections import defaultdict
out, key, source = sys.argv[1], sys.argv[2], sys.argv[3]
vals = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(os.path.join(out, "pmc*", "*counter_collection.csv"))):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            n = row.get("Kernel_Name", "")
            if n.startswith("clx_k_"):
                vals[n][row["Counter_Name"]].append(float(row["Counter_Value"]))
mean = lambda v: sum(v) / len(v)
kern, insts = {}, {}
for n, cs in vals.items():
    if "FETCH_SIZE" in cs:
        kern[n] = int((2 * mean(cs["FETCH_SIZE"]) + mean(cs.get("WRITE_SIZE", [0.0]))) * 1024)
    if "SQ_INSTS_VALU" in cs:
        insts[n] = {"valu": int(mean(cs["SQ_INSTS_VALU"])), "salu": int(mean(cs["SQ_INSTS_SALU"])), "lds": int(mean(cs["SQ_INSTS_LDS"])),
                    "waves": int(mean(cs["SQ_WAVES"]))}
line = json.loads(open(os.path.join(out, "bench_line.json")).read())
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
p = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "pmc_traffic.json")
j = json.load(open(p))
j[key] = {"path_bytes": sum(kern.values()), "kernels": kern, "insts": insts, "algorithmic_bytes": line["roofline"]["algorithmic_bytes_per_launch"],
          "samples_per_launch": line["config"]["samples_per_step"], "source": source, "kernel_src_sha16": bench.kernel_source_sha16()}
json.dump(j, open(p, "w"), indent=1)
print(key, j[key]["path_bytes"], round(j[key]["path_bytes"] / j[key]["algorithmic_bytes"], 3), {k: v["valu"] for k, v in insts.items()})
