set -e
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04r
mkdir -p $OUT
rm -f $OUT/ab.txt
timeout -k 10 900 python -m pytest tests/test_tracking.py tests/test_gpu_slam_loops.py tests/test_gpu_round3.py tests/test_gpu_round4.py -x -q -m gpu -k "not full_size and not k_keyframe" > $OUT/pytest.log 2>&1 || { tail -40 $OUT/pytest.log; exit 1; }
tail -1 $OUT/pytest.log
for rep in 1 2 3; do
for v in pose4 product; do
  if [ $v = product ]; then unset GS2D_LIB_PATH; else export GS2D_LIB_PATH=$PWD/scripts/dev/variants/lib$v.so; fi
  timeout -k 10 200 python scripts/dev/stage_ms.py 1 --workload tracking >> $OUT/ab.txt 2>&1
done
done
cat $OUT/ab.txt
