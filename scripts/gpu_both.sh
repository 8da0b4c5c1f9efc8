set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m gaus_slam_amd.build 2>&1 | tail -1
make -s -C oracle
timeout -k 10 500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tee gpurun_out/bench_latest.log | tail -3
