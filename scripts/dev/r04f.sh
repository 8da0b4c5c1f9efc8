set -e
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04f
mkdir -p $OUT
rm -f $OUT/ab_bin.txt
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "scannetpp or replica or capacity or guess or culled or 1200" > $OUT/pytest_big.log 2>&1 || { tail -40 $OUT/pytest_big.log; exit 1; }
tail -3 $OUT/pytest_big.log
for w in scannetpp scannetpp_ref replica; do
  for it in 4096 8192 16384; do
    echo "$w items=$it" >> $OUT/ab_bin.txt
    GS2D_BIN_ITEMS_FORCE=$it timeout -k 10 300 python scripts/dev/stage_ms.py 1 --workload $w >> $OUT/ab_bin.txt 2>&1
  done
done
cat $OUT/ab_bin.txt
