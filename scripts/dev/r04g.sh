set -e
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04g
mkdir -p $OUT
export TMPDIR=/tmp
for w in scannetpp scannetpp_ref; do
D=$GRAFT_REPO_ROOT/$OUT/trace_$w
(cd /tmp && timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 $GRAFT_REPO_ROOT/bench.py --workload $w --steps 20 --warmup 5 --prewarm-steps 50 --no-cpu-baseline --no-extra-legs > $D.log 2>&1) || true
python3 - <<PY
import csv,glob,collections
rows=collections.defaultdict(list)
for f in glob.glob("$D/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"]
        import re
        m=re.search(r"::(\w+)\s*(?:<|\()", n)
        rows[(m.group(1) if m else n[:30], r["Grid_Size_X"], r["LDS_Block_Size"])].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
print("$w")
for k,v in sorted(rows.items(), key=lambda kv:-sum(kv[1]))[:12]:
    print(f"  {k[0]:30s} grid {k[1]:>8s} lds {k[2]:>6s} calls {len(v):4d} avg {sum(v)/len(v):8.2f} us")
PY
done
