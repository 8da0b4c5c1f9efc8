#!/usr/bin/env python3
"""Condenses a rocprofv3 output directory (kernel trace + FETCH_SIZE / WRITE_SIZE passes) into a small text/JSON
summary that is committed under profiles/.  FETCH_SIZE is doubled for wide coalesced reads as the gfx950 guide
prescribes (MI355X_MICROARCH.md, HBM section); both raw and corrected numbers are printed."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

d = sys.argv[1]


def short(n):
    for k in ("blend_fwd_kernel", "blend_bwd_kernel", "preprocess_fwd_kernel", "preprocess_bwd_pose_kernel", "preprocess_bwd_kernel",
              "radix_hist_kernel", "radix_scatter_kernel", "scan_reduce_kernel", "scan_blocksums_kernel", "scan_apply_kernel",
              "duplicate_kernel", "tile_ranges_kernel", "tile_depth_sort_dev_kernel", "tile_depth_sort_kernel", "bin_hist_dev_kernel",
              "bin_row_scan_dev_kernel", "bin_scatter_dev_kernel", "bin_hist_kernel", "bin_row_scan_kernel", "bin_scatter_kernel",
              "sum_frames_kernel", "mark_visible_kernel", "knn_kernel", "det_reduce_kernel", "det_inverse_kernel",
              "slam_loss", "adam"):
        if k in n:
            # the tracking backward runs the POSE instantiation of blend_bwd_kernel (fourth template argument true)
            if k == "blend_bwd_kernel":
                targs = n.split("blend_bwd_kernel<", 1)[1].split(">", 1)[0].replace(" ", "").split(",") if "blend_bwd_kernel<" in n else []
                if len(targs) >= 4 and targs[3] == "true":
                    return "blend_bwd_kernel<POSE>"
            return k
    return n.split("<")[0].split("(")[0][:60]


def kernel_trace(sub):
    rows = defaultdict(list)
    for f in glob.glob(os.path.join(d, sub, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return rows


def counter(sub, name):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(d, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == name:
                acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return acc


tr = kernel_trace("trace")
print(f"{'kernel':28s} {'calls':>6s} {'avg_us':>10s} {'total_ms':>10s}")
tot = sum(sum(v) for v in tr.values())
for k, v in sorted(tr.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k:28s} {len(v):6d} {sum(v)/len(v):10.2f} {sum(v)/1e3:10.3f}  {100*sum(v)/tot:5.1f}%")
fe, wr = counter("pmc_fetch", "FETCH_SIZE"), counter("pmc_write", "WRITE_SIZE")
out = {}
print("\nHBM traffic per launch (counters in KiB as reported; FETCH_SIZE x2 = gfx950 correction, 128-byte requests counted as 64; "
      "total in MB = 1e6 bytes, the unit of bench.py's roofline.traffic)")
for k in sorted(set(fe) | set(wr)):
    f = sum(fe[k]) / len(fe[k]) if fe.get(k) else 0.0
    w = sum(wr[k]) / len(wr[k]) if wr.get(k) else 0.0
    print(f"{k:28s} FETCH_SIZE {f:12.1f} KiB  WRITE_SIZE {w:12.1f} KiB  corrected total {(2*f+w)*1024/1e6:10.2f} MB")
    out[k] = {"fetch_kb_raw": f, "write_kb": w, "bytes_corrected": (2 * f + w) * 1024}
json.dump({"kernel_avg_us": {k: sum(v) / len(v) for k, v in tr.items()}, "traffic": out},
          open(os.path.join(d, "summary.json"), "w"), indent=1)
