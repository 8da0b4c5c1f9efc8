set -e
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04c
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1 || { tail -40 $OUT/pytest_gpu.log; exit 1; }
tail -3 $OUT/pytest_gpu.log
for rep in 1 2; do
  GS2D_LAUNCH_AHEAD=0 timeout -k 10 200 python scripts/dev/stage_ms.py 1 >> $OUT/ab_ahead.txt 2>&1
  timeout -k 10 200 python scripts/dev/stage_ms.py 1 >> $OUT/ab_ahead.txt 2>&1
done
cat $OUT/ab_ahead.txt
