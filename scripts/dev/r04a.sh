set -e
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04a
mkdir -p $OUT
timeout -k 10 500 python -m pytest tests/test_tracking.py tests/test_gpu_slam_loops.py -x -q -m gpu -s > $OUT/pytest_tracking.log 2>&1 || { tail -30 $OUT/pytest_tracking.log; exit 1; }
tail -5 $OUT/pytest_tracking.log
grep "pose-only vs" $OUT/pytest_tracking.log || true
for rep in 1 2; do
  GS2D_LIB_PATH=$PWD/scripts/dev/variants/libr03.so timeout -k 10 200 python scripts/dev/stage_ms.py 1 --workload tracking >> $OUT/ab_tracking.txt 2>&1
  timeout -k 10 200 python scripts/dev/stage_ms.py 1 --workload tracking >> $OUT/ab_tracking.txt 2>&1
done
cat $OUT/ab_tracking.txt
