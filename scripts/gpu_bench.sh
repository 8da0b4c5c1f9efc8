set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout -k 10 400 python bench.py --steps 20 --warmup 5 2>&1 | tee gpurun_out/bench_first.log | tail -5
