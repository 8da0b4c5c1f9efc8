# LDS counters of the blend kernels (rocprofv3 --pmc, own pass): bank conflicts and LDS activity.  usage on the GPU box:
#   bash scripts/gpu_pmc_lds.sh <tag>
set -e
cd $GRAFT_REPO_ROOT
TAG=${1:-lds}
mkdir -p gpurun_out/prof_$TAG
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG/lds -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --prewarm-steps 20 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG/lds.log 2>&1) || true
python3 - <<PY
import csv, glob, collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/prof_$TAG/lds/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        for k in ('blend_fwd_kernel','blend_bwd_kernel','tile_depth_sort_kernel'):
            if k in r['Kernel_Name']: acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items():
    print(k, {c: round(sum(x)/len(x)) for c,x in v.items()})
PY
tail -3 gpurun_out/prof_$TAG/lds.log
