# Dev: the driver's command repeated with the host's wait for num_rendered sleeping (default) or spinning (GS2D_NAP_WAIT=0).
# usage (GPU box): bash scripts/dev/nap_ab.sh [reps=10] -> one line per run: nap value ms cold refbin host_cpu
cd $GRAFT_REPO_ROOT
REPS=${1:-10}
for i in $(seq $REPS); do
  for N in 1 0; do
    GS2D_NAP_WAIT=$N python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print($N, d['value'], d['ms_per_step'], d['config']['cold_value'], d['config']['reference_binning_value'], d['config']['host_cpu_fraction'])"
  done
done
