#!/usr/bin/env python3
"""Basic blocks of one kernel in a hipcc -S listing: line, label, VALU / SALU / LDS / VMEM counts, where it branches.
usage: tools/isa_blocks.py file.s kernel [first_label last_label]"""
import re, sys
def blocks(path, kernel):
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(kernel + ":"))
    out, cur = [], None
    for i in range(start, len(lines)):
        l = lines[i]
        if l.startswith(".Lfunc_end"): break
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m or cur is None:
            cur = {"line": i + 1, "label": m.group(1) if m else kernel, "v": 0, "s": 0, "d": 0, "g": 0, "br": [], "text": []}
            out.append(cur)
            if m: continue
        t = l.strip()
        if not t or t.startswith(";") or t.startswith("."): continue
        cur["text"].append(t)
        op = t.split()[0]
        if op.startswith("v_"): cur["v"] += 1
        elif op.startswith("ds_"): cur["d"] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")): cur["g"] += 1
        elif op.startswith("s_"):
            cur["s"] += 1
            if op.startswith(("s_cbranch", "s_branch")): cur["br"].append(op[2:] + ">" + t.split()[-1].replace(".LBB", ""))
    return out
if __name__ == "__main__":
    bl = blocks(sys.argv[1], sys.argv[2])
    lo = sys.argv[3] if len(sys.argv) > 3 else None
    hi = sys.argv[4] if len(sys.argv) > 4 else None
    on = lo is None
    for b in bl:
        if b["label"] == lo: on = True
        if on: print("%7d %-14s v%4d s%4d ds%3d vm%3d  %s" % (b["line"], b["label"], b["v"], b["s"], b["d"], b["g"], " ".join(b["br"])))
        if b["label"] == hi: on = False
