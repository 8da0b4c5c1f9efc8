#!/bin/bash
# Dev tool: run scripts/dev/fetch_size_probe under rocprofv3 --pmc FETCH_SIZE and print the counter beside the known bytes.
# usage (GPU box): bash scripts/dev/fetch_size_probe.sh > gpurun_out/fetch_size_probe.txt
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/fetchprobe
rm -rf $OUT; mkdir -p $OUT
./scripts/dev/fetch_size_probe | tail -2
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc -- ./scripts/dev/fetch_size_probe > $OUT/run.log 2>&1
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE":
            acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
known = {"stream_read_kernel": (2 * 2**30, "2 GiB streamed (coalesced float4 per lane)"),
         "gather80_kernel": (8 * 2**20 * 80, "8 Mi records x 80 B algorithmic; 1074 MB as 64-B lines, 1611 MB as 128-B lines; + 34 MB of indices")}
for k, v in acc.items():
    kb = sum(v) / len(v)
    if k in known:
        print(f"{k}: FETCH_SIZE = {kb:.0f} KB = {kb * 1024 / 1e6:.1f} MB per launch; known: {known[k][1]}; counter / algorithmic bytes = {kb * 1024 / known[k][0]:.3f}")
PY
