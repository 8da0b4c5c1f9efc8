set -e
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04q
mkdir -p $OUT
timeout -k 10 1000 python -m pytest tests -x -q -m gpu --deselect tests/test_host.py > $OUT/pytest_gpu.log 2>&1 || { tail -60 $OUT/pytest_gpu.log; exit 1; }
tail -3 $OUT/pytest_gpu.log
