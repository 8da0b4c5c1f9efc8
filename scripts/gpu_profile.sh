# rocprofv3 kernel trace + HBM traffic counters for the bench workload.  Usage: bash scripts/gpu_profile.sh <tag>
set -e
cd $GRAFT_REPO_ROOT
TAG=${1:-r01}
mkdir -p gpurun_out/prof_$TAG
python -m gaus_slam_amd.build >/dev/null
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG/trace -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/prof_$TAG/bench_under_profiler.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_$TAG/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof_$TAG/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_$TAG/pmc_write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof_$TAG/pmc_write.log 2>&1
find gpurun_out/prof_$TAG -name "*.csv" | head -20
python3 scripts/summarize_profile.py gpurun_out/prof_$TAG > gpurun_out/prof_$TAG/summary.txt 2>&1 || true
cat gpurun_out/prof_$TAG/summary.txt | head -60
