set -e
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04b
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_round3.py -x -q -m gpu -s -k "knife or flipped or full_size" > $OUT/pytest_f64.log 2>&1 || { tail -40 $OUT/pytest_f64.log; exit 1; }
grep -E "vs float64|flipped|knife|passed|failed" $OUT/pytest_f64.log
