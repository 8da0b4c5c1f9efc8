#!/usr/bin/env python3
"""Dev tool: per-kernel launch count / mean / total duration from a rocprofv3 --kernel-trace directory, restricted to the
last `steps` fraction of the run (skips warm-up launches).  usage: kernel_times.py <dir> [tail_fraction=0.3]"""
import csv, glob, os, re, sys
from collections import defaultdict
d = sys.argv[1]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
rows = rows[int(len(rows) * (1 - frac)):]
acc = defaultdict(list)
for s, e, n in rows:
    m = re.search(r"::(\w+)\s*(?:<|\()", n)
    k = m.group(1) if m else n.split("(")[0]
    for key in ("blend_fwd_kernel", "blend_bwd_kernel"):
        if key in n:
            k = key + ("<batch>" if "ELb1EEEv" in n or "true>" in n.split("(")[0][-8:] else "")
    acc[k[:70]].append((e - s) / 1e3)
span = (rows[-1][1] - rows[0][0]) / 1e3
busy = sum(e - s for s, e, _ in rows) / 1e3
print(f"window {span:.0f} us, kernels busy {busy:.0f} us ({busy / span:.1%}), {len(rows)} launches")
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k:72s} n={len(v):5d} mean {sum(v) / len(v):8.1f} us total {sum(v):9.0f} us")
