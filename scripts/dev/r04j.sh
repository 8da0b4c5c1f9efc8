set -e
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04j
mkdir -p $OUT
rm -f $OUT/ab.txt
for v in product ng8nc ng4pad ng4pad448 ng4nc; do
  if [ $v = product ]; then unset GS2D_LIB_PATH; else export GS2D_LIB_PATH=$PWD/scripts/dev/variants/lib$v.so; fi
  timeout -k 10 200 python scripts/dev/stage_ms.py 1 >> $OUT/ab.txt 2>&1
  timeout -k 10 200 python scripts/dev/stage_ms.py 1 --workload tracking >> $OUT/ab.txt 2>&1
done
cat $OUT/ab.txt
