set -e
cd $GRAFT_REPO_ROOT
TAG=${1:-sq}
mkdir -p gpurun_out/prof_$TAG
export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d gpurun_out/prof_$TAG/sq -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof_$TAG/sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/prof_$TAG/sq2 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof_$TAG/sq2.log 2>&1 || true
python3 - <<'PY'
import csv, glob, collections, os
tag=os.environ.get('TAG','sq')
for sub in ('sq','sq2'):
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f'gpurun_out/prof_{tag}/{sub}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            n=r['Kernel_Name']
            for k in ('blend_fwd_kernel','blend_bwd_kernel','tile_depth_sort_kernel'):
                if k in n: acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in acc.items():
        print(k, {c: round(sum(x)/len(x)) for c,x in v.items()})
PY
