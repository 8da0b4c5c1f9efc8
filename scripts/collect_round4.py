#!/usr/bin/env python3
"""Copies the artifacts scripts/gpu_round4_artifacts.sh left under gpurun_out/<tag>/ into profiles/ under their round-4 names.
Every JSON kept carries the hash of the kernel sources it was measured with (bench.py: build.source_hash; PMC summaries:
source_hash); the auxiliary probes (kNN, deterministic backward, cull tightness) are stamped with the hash of the session's
bench JSONs after checking that all of those agree.  usage: collect_round4.py [tag=r04_final]"""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04_final"
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")
hashes = set()


def put_json(path, name, stamp=None):
    text = open(path).read().strip()
    try:
        d = json.loads(text)
    except json.JSONDecodeError:
        d = json.loads(text.splitlines()[-1])  # a log with the JSON as its last line
    if stamp is not None:
        d["source_hash"] = stamp
    h = (d.get("build") or {}).get("source_hash") or d.get("source_hash")
    hashes.add(h)
    with open(os.path.join(dst, name), "w") as f:
        f.write(json.dumps(d) + "\n")
    print(name, h)


for w in ("op", "b200k", "replica", "scannetpp", "scannetpp_ref", "tracking", "mapping", "tracking_replica", "mapping_replica"):
    d = os.path.join(src, "prof_" + w)
    if not os.path.isdir(d):
        continue
    suffix = "final" if w == "op" else w
    shutil.copy(os.path.join(d, "summary.txt"), os.path.join(dst, f"rocprof_r04_{suffix}_summary.txt"))
    ks = glob.glob(os.path.join(d, "trace", "**", "*kernel_stats.csv"), recursive=True)
    if ks:
        shutil.copy(ks[0], os.path.join(dst, f"rocprof_r04_{suffix}_kernel_stats.csv"))
    stem = "" if w == "op" else "_" + w
    put_json(os.path.join(d, "pmc_traffic.json"), f"pmc_traffic{stem}_r04.json")
    put_json(os.path.join(d, "pmc_valu.json"), f"pmc_valu{stem}_r04.json")
for p in sorted(glob.glob(os.path.join(src, "bench_*.json"))):
    put_json(p, "bench_r04_" + os.path.basename(p)[len("bench_"):])
assert len(hashes) == 1, f"artifacts of more than one build: {hashes}"
h = next(iter(hashes))
for name in ("knn_bench", "det_bench", "cull_tightness"):
    p = os.path.join(src, name + ".json")
    if os.path.exists(p):
        put_json(p, f"{name}_r04.json", stamp=h)
with open(os.path.join(dst, "timeline_r04.txt"), "w") as f:
    for name, title in (("timeline_k1.txt", "one keyframe per step, launch-ahead forward (default)"),
                        ("timeline_k1_ahead0.txt", "one keyframe per step, GS2D_LAUNCH_AHEAD=0 (the round-3 order: duplicate, host wait, the rest)"),
                        ("timeline_k4.txt", "four keyframes per step in one batched call")):
        p = os.path.join(src, name)
        if os.path.exists(p):
            f.write(f"# {title}\n" + open(p).read() + "\n")
lds = glob.glob(os.path.join(src, "lds", "**", "*counter_collection.csv"), recursive=True)
if lds:
    import collections, csv
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(lds[0])):
        for k in ("blend_fwd_kernel", "blend_bwd_kernel"):
            if k in r["Kernel_Name"]:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    with open(os.path.join(dst, "pmc_lds_r04.txt"), "w") as f:
        f.write(f"# rocprofv3 --pmc SQ_LDS_* of `bench.py --steps 3` (averages per launch), kernel sources {h}\n")
        for k, cs in acc.items():
            f.write(k + ": " + "  ".join(f"{c} {sum(v) / len(v):.0f}" for c, v in sorted(cs.items())) + "\n")
print("source hash of the set:", h)
