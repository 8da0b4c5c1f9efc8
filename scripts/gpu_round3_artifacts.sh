# Produces the measured artifacts of a round under gpurun_out/<tag>/ (copied into profiles/ by hand afterwards).
# usage on the GPU box: bash scripts/gpu_round3_artifacts.sh <tag> <part>   (part 1: headline bench + rocprof + PMC; part 2: other workloads; part 3: headline bench again -- it quotes the PMC files part 1 produced, once they are in profiles/ -- and the two-rank rehearsal)
set -e
cd $GRAFT_REPO_ROOT
TAG=${1:-r03_final}
PART=${2:-1}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
if [ "$PART" = "1" ]; then
  timeout -k 10 400 python bench.py --json-out $OUT/bench_final.json > $OUT/bench_final.log 2>&1 || true
  echo "bench done"; tail -c 400 $OUT/bench_final.log
  (cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --prewarm-steps 100 --no-cpu-baseline --no-extra-legs > $GRAFT_REPO_ROOT/$OUT/trace.log 2>&1) || true
  echo "trace done"
  (cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_fetch -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --prewarm-steps 20 --no-cpu-baseline --no-extra-legs > $GRAFT_REPO_ROOT/$OUT/pmc_fetch.log 2>&1) || true
  (cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_write -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --prewarm-steps 20 --no-cpu-baseline --no-extra-legs > $GRAFT_REPO_ROOT/$OUT/pmc_write.log 2>&1) || true
  echo "traffic done"
  (cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $GRAFT_REPO_ROOT/$OUT/sq -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --prewarm-steps 20 --no-cpu-baseline --no-extra-legs > $GRAFT_REPO_ROOT/$OUT/sq.log 2>&1) || true
  (cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --output-format csv -d $GRAFT_REPO_ROOT/$OUT/sq2 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --prewarm-steps 20 --no-cpu-baseline --no-extra-legs > $GRAFT_REPO_ROOT/$OUT/sq2.log 2>&1) || true
  echo "sq done"
  python3 scripts/summarize_profile.py $OUT > $OUT/summary.txt 2>&1 || true
  python3 scripts/make_pmc_json.py $OUT > $OUT/make_pmc.log 2>&1 || true
  head -30 $OUT/summary.txt
elif [ "$PART" = "2" ]; then
  for w in b200k replica scannetpp scannetpp_ref tracking mapping; do
    timeout -k 10 300 python bench.py --workload $w --steps 30 --warmup 5 --no-cpu-baseline --no-extra-legs --json-out $OUT/bench_$w.json > $OUT/bench_$w.log 2>&1 || true
    echo "$w: $(python3 -c "import json;d=json.load(open('$OUT/bench_$w.json'));print(d['value'],d['unit'],d['ms_per_step'])" 2>/dev/null)"
  done
  for k in 2 4 8; do
    timeout -k 10 300 python bench.py --keyframes-per-gpu $k --steps 30 --warmup 5 --no-cpu-baseline --no-extra-legs --json-out $OUT/bench_kpg$k.json > $OUT/bench_kpg$k.log 2>&1 || true
    echo "kpg$k: $(python3 -c "import json;d=json.load(open('$OUT/bench_kpg$k.json'));print(d['value'],d['unit'],d['ms_per_step'])" 2>/dev/null)"
  done
  timeout -k 10 300 python bench.py --keyframes-per-gpu 4 --no-batch --steps 30 --warmup 5 --no-cpu-baseline --no-extra-legs --json-out $OUT/bench_kpg4_nobatch.json > $OUT/bench_kpg4_nobatch.log 2>&1 || true
  timeout -k 10 200 python scripts/dev/batch_host_profile.py 1 2>&1 | grep -E "K=|timeline" > $OUT/timeline_k1.txt || true
  timeout -k 10 200 python scripts/dev/batch_host_profile.py 4 2>&1 | grep -E "K=|timeline" > $OUT/timeline_k4.txt || true
  (cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $GRAFT_REPO_ROOT/$OUT/lds -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --prewarm-steps 20 --no-cpu-baseline --no-extra-legs > $GRAFT_REPO_ROOT/$OUT/lds.log 2>&1) || true
  timeout -k 10 300 python bench.py --adam fused --steps 30 --warmup 5 --no-cpu-baseline --no-extra-legs --json-out $OUT/bench_adam_fused.json > $OUT/bench_adam.log 2>&1 || true
  timeout -k 10 200 python scripts/dev/knn_bench.py > $OUT/knn_bench.json 2>&1 || true
  timeout -k 10 200 python scripts/dev/det_bench.py > $OUT/det_bench.json 2>&1 || true
  GS2D_LIB_PATH=$PWD/scripts/dev/variants/libprof.so timeout -k 10 200 python scripts/dev/wave_profile.py > $OUT/wave_profile.txt 2>&1 || true
  timeout -k 10 300 python scripts/dev/cull_tightness.py > $OUT/cull_tightness.json 2>&1 || true
  tail -1 $OUT/knn_bench.json; tail -1 $OUT/det_bench.json; tail -1 $OUT/cull_tightness.json
fi
if [ "$PART" = "3" ]; then
  timeout -k 10 400 python bench.py --json-out $OUT/bench_final.json > $OUT/bench_final.log 2>&1 || true
  tail -c 300 $OUT/bench_final.log
  timeout -k 10 300 python bench.py --gpus 2 --backend gloo --share-device --steps 5 --warmup 2 --no-cpu-baseline --json-out $OUT/bench_rehearsal_2ranks_gloo_one_gpu.json > $OUT/bench_rehearsal.log 2>&1 || true
  tail -c 600 $OUT/bench_rehearsal.log
fi
