#!/usr/bin/env python3
"""Turns a profiling directory made by scripts/gpu_round2_artifacts.sh (rocprofv3 PMC passes of `bench.py`) into the two
small JSON files bench.py reads: pmc_traffic (HBM bytes per launch per stage, FETCH_SIZE doubled as the gfx950 guide
prescribes) and pmc_valu (SQ_INSTS_VALU / SQ_INSTS_SALU per launch of the blend kernels).  usage: make_pmc_json.py <dir>"""
import collections
import csv
import glob
import json
import os
import re
import sys

d = sys.argv[1]
STAGE = {"blend_fwd_kernel": "blend_fwd", "blend_bwd_kernel": "blend_bwd", "preprocess_fwd_kernel": "preprocess",
         "preprocess_bwd_kernel": "preprocess_bwd", "preprocess_bwd_pose_kernel": "preprocess_bwd",
         "bin_hist_kernel": "sort", "bin_row_scan_kernel": "sort", "bin_scatter_kernel": "sort", "tile_depth_sort_kernel": "sort",
         "bin_hist_dev_kernel": "sort", "bin_row_scan_dev_kernel": "sort", "bin_scatter_dev_kernel": "sort", "tile_depth_sort_dev_kernel": "sort",
         "radix_hist_kernel": "sort", "radix_scatter_kernel": "sort", "tile_ranges_kernel": "sort",
         "scan_reduce_kernel": "sort", "scan_apply_kernel": "sort", "duplicate_kernel": "duplicate"}
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaus_slam_amd import build as _gs_build  # noqa: E402  (the hash of the kernel sources these counters belong to)
SOURCE_HASH = _gs_build.source_hash()


CALLS = {}  # launches seen per kernel (the largest of the passes)


def counters(sub):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"::(\w+)\s*(?:<|\()", r["Kernel_Name"])  # "void (anonymous namespace)::blend_bwd_kernel<true, ...>(int, ..."
            name = m.group(1) if m else ""
            if name in STAGE:
                acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    CALLS.update({k: max(CALLS.get(k, 0), max(len(v) for v in cs.values())) for k, cs in acc.items()})
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}


fe, wr, sq, sq2 = counters("pmc_fetch"), counters("pmc_write"), counters("sq"), counters("sq2")
traffic = collections.defaultdict(float)
# a stage is the kernels the timed steps launch: variants that ran once or twice in the set-up (the host-count forms of the
# binning kernels under a debug / instance-count forward) are left out
stage_calls = collections.defaultdict(int)
for k, st in STAGE.items():
    stage_calls[st] = max(stage_calls[st], CALLS.get(k, 0))
for k, st in STAGE.items():
    if CALLS.get(k, 0) * 2 < stage_calls[st]:
        continue
    f = fe.get(k, {}).get("FETCH_SIZE", 0.0)
    w = wr.get(k, {}).get("WRITE_SIZE", 0.0)
    traffic[st] += (2 * f + w) * 1024  # KB as reported -> bytes; FETCH_SIZE counts 64 B per 128-B request on gfx950
out_t = {k: int(v) for k, v in traffic.items()}
out_t["_note"] = ("HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes of `bench.py --steps 3`), "
                  "FETCH_SIZE doubled per the gfx950 guide (MI355X_MICROARCH.md, HBM) -- verified for 80-byte gathers as well, profiles/fetch_size_probe_r03.txt; "
                  "sort = bin_hist + bin_row_scan + bin_scatter (+ tile_depth_sort when it runs as a kernel of its own; at the headline size it is a phase of blend_fwd)")
out_v = {}
for k in ("blend_fwd_kernel", "blend_bwd_kernel"):
    if k in sq:
        out_v[STAGE[k]] = {"valu_wave_insts": int(sq[k].get("SQ_INSTS_VALU", 0)), "salu_wave_insts": int(sq[k].get("SQ_INSTS_SALU", 0)),
                           "wave_quad_cycles": int(sq[k].get("SQ_WAVE_CYCLES", 0)), "wait_any": int(sq[k].get("SQ_WAIT_ANY", 0)),
                           "wait_inst_any": int(sq[k].get("SQ_WAIT_INST_ANY", 0)), "active_inst_any": int(sq[k].get("SQ_ACTIVE_INST_ANY", 0)),
                           "lds_insts": int(sq2.get(k, {}).get("SQ_INSTS_LDS", 0)), "waves": int(sq2.get(k, {}).get("SQ_WAVES", 0)),
                           "kernel_cycles": int(sq2.get(k, {}).get("GRBM_GUI_ACTIVE", 0) / 8)}
out_v["_note"] = ("rocprofv3 --pmc SQ_* passes of `bench.py --steps 3` (averages per launch; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* are "
                  "quad-cycles summed over waves; kernel_cycles = GRBM_GUI_ACTIVE / 8)")
out_t["source_hash"] = SOURCE_HASH
out_v["source_hash"] = SOURCE_HASH
json.dump(out_t, open(os.path.join(d, "pmc_traffic.json"), "w"), indent=1)
json.dump(out_v, open(os.path.join(d, "pmc_valu.json"), "w"), indent=1)
print(json.dumps(out_t, indent=1))
print(json.dumps(out_v, indent=1))
