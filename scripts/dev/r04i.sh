set -e
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04i
mkdir -p $OUT
rm -f $OUT/ab_ng.txt
# correctness of the eight-queue build first: the backward parity tests + tracking + cull bits
GS2D_LIB_PATH=$PWD/scripts/dev/variants/libng8.so timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "not host and not build_info" > $OUT/pytest_ng8.log 2>&1 || { tail -40 $OUT/pytest_ng8.log; exit 1; }
tail -3 $OUT/pytest_ng8.log
for rep in 1 2 3; do
  for v in product ng8; do
    if [ $v = product ]; then unset GS2D_LIB_PATH; else export GS2D_LIB_PATH=$PWD/scripts/dev/variants/lib$v.so; fi
    timeout -k 10 200 python scripts/dev/stage_ms.py 1 >> $OUT/ab_ng.txt 2>&1
  done
done
for rep in 1 2; do
  for v in product ng8; do
    if [ $v = product ]; then unset GS2D_LIB_PATH; else export GS2D_LIB_PATH=$PWD/scripts/dev/variants/lib$v.so; fi
    timeout -k 10 200 python scripts/dev/stage_ms.py 1 --workload tracking >> $OUT/ab_ng.txt 2>&1
  done
done
cat $OUT/ab_ng.txt
