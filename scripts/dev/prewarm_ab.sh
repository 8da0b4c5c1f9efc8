# Dev: the driver's command (bench.py --gpus 1 --steps 20 --warmup 5) repeated with different pre-warm durations.
# usage (GPU box): bash scripts/dev/prewarm_ab.sh [reps=8] -> one line per run: seconds value ms cold refbin host_cpu
cd $GRAFT_REPO_ROOT
REPS=${1:-8}
for i in $(seq $REPS); do
  for S in 1.6 0.4 0.8; do
    python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --prewarm-seconds $S 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print($S, d['value'], d['ms_per_step'], d['config']['cold_value'], d['config']['reference_binning_value'], d['config']['host_cpu_fraction'], d['prewarm_steps'])"
  done
done
