# Produces the measured artifacts of round 4 under gpurun_out/<tag>/ (copied into profiles/ afterwards by scripts/collect_round4.sh).
# usage on the GPU box: bash scripts/gpu_round4_artifacts.sh <tag> <part>
#   part prof:<workload>   kernel trace + FETCH_SIZE / WRITE_SIZE / SQ passes of `bench.py --workload <workload>` (op = headline)
#   part bench             one bench JSON per workload + keyframes-per-GPU runs
#   part misc              timelines, LDS counters, det / knn / wave-profile / cull-tightness probes
#   part final             headline bench (quotes the PMC files once they are in profiles/) + two-rank rehearsal
set -e
cd $GRAFT_REPO_ROOT
TAG=${1:-r04_final}
PART=${2:-prof:op}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
B="--no-cpu-baseline --no-extra-legs"
prof() {  # prof <workload>
  local w=$1; local D=$GRAFT_REPO_ROOT/$OUT/prof_$w; local WL=""
  [ "$w" != "op" ] && WL="--workload $w"
  mkdir -p $D
  (cd /tmp && timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace -- python3 $GRAFT_REPO_ROOT/bench.py $WL --steps 20 --warmup 5 --prewarm-steps 100 $B > $D/trace.log 2>&1) || true
  for c in "pmc_fetch FETCH_SIZE" "pmc_write WRITE_SIZE" "sq SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "sq2 SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE"; do
    set -- $c; local sub=$1; shift
    (cd /tmp && timeout -k 10 240 rocprofv3 --kernel-trace --pmc $@ --output-format csv -d $D/$sub -- python3 $GRAFT_REPO_ROOT/bench.py $WL --steps 3 --warmup 1 --prewarm-steps 20 $B > $D/$sub.log 2>&1) || true
  done
  python3 scripts/summarize_profile.py $OUT/prof_$w > $OUT/prof_$w/summary.txt 2>&1 || true
  python3 scripts/make_pmc_json.py $OUT/prof_$w > $OUT/prof_$w/make_pmc.log 2>&1 || true
  echo "== $w"; head -12 $OUT/prof_$w/summary.txt
}
case "$PART" in
prof:*)
  for w in $(echo ${PART#prof:} | tr ',' ' '); do prof $w; done ;;
bench)
  for w in b200k replica scannetpp scannetpp_ref tracking mapping tracking_replica mapping_replica; do
    timeout -k 10 300 python bench.py --workload $w --steps 30 --warmup 5 $B --json-out $OUT/bench_$w.json > $OUT/bench_$w.log 2>&1 || true
    echo "$w: $(python3 -c "import json;d=json.load(open('$OUT/bench_$w.json'));print(d['value'],d['unit'],d['ms_per_step'])" 2>/dev/null)"
  done
  for k in 2 4 8; do
    timeout -k 10 300 python bench.py --keyframes-per-gpu $k --steps 30 --warmup 5 $B --json-out $OUT/bench_kpg$k.json > $OUT/bench_kpg$k.log 2>&1 || true
    echo "kpg$k: $(python3 -c "import json;d=json.load(open('$OUT/bench_kpg$k.json'));print(d['value'],d['unit'],d['ms_per_step'])" 2>/dev/null)"
  done
  timeout -k 10 300 python bench.py --keyframes-per-gpu 4 --no-batch --steps 30 --warmup 5 $B --json-out $OUT/bench_kpg4_nobatch.json > $OUT/bench_kpg4_nobatch.log 2>&1 || true
  timeout -k 10 300 python bench.py --adam fused --steps 30 --warmup 5 $B --json-out $OUT/bench_adam_fused.json > $OUT/bench_adam.log 2>&1 || true ;;
misc)
  timeout -k 10 200 python scripts/dev/batch_host_profile.py 1 2>&1 | grep -E "K=|timeline" > $OUT/timeline_k1.txt || true
  GS2D_LAUNCH_AHEAD=0 timeout -k 10 200 python scripts/dev/batch_host_profile.py 1 2>&1 | grep -E "K=|timeline" > $OUT/timeline_k1_ahead0.txt || true
  timeout -k 10 200 python scripts/dev/batch_host_profile.py 4 2>&1 | grep -E "K=|timeline" > $OUT/timeline_k4.txt || true
  (cd /tmp && timeout -k 10 240 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $GRAFT_REPO_ROOT/$OUT/lds -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --prewarm-steps 20 $B > $GRAFT_REPO_ROOT/$OUT/lds.log 2>&1) || true
  timeout -k 10 200 python scripts/dev/knn_bench.py > $OUT/knn_bench.json 2>&1 || true
  timeout -k 10 200 python scripts/dev/det_bench.py > $OUT/det_bench.json 2>&1 || true
  timeout -k 10 300 python scripts/dev/cull_tightness.py > $OUT/cull_tightness.json 2>&1 || true
  tail -1 $OUT/knn_bench.json; tail -1 $OUT/det_bench.json; tail -1 $OUT/cull_tightness.json; cat $OUT/timeline_k1.txt ;;
final)
  timeout -k 10 400 python bench.py --json-out $OUT/bench_final.json > $OUT/bench_final.log 2>&1 || true
  tail -c 300 $OUT/bench_final.log
  timeout -k 10 300 python bench.py --gpus 2 --backend gloo --share-device --steps 5 --warmup 2 --no-cpu-baseline --json-out $OUT/bench_rehearsal_2ranks_gloo_one_gpu.json > $OUT/bench_rehearsal.log 2>&1 || true
  tail -c 400 $OUT/bench_rehearsal.log ;;
esac
