set -e
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04d
mkdir -p $OUT
rm -f $OUT/ab_nap.txt
for rep in 1 2 3; do
  echo "nap=0" >> $OUT/ab_nap.txt; GS2D_NAP_WAIT=0 timeout -k 10 200 python scripts/dev/stage_ms.py 1 >> $OUT/ab_nap.txt 2>&1
  echo "nap=1" >> $OUT/ab_nap.txt; timeout -k 10 200 python scripts/dev/stage_ms.py 1 >> $OUT/ab_nap.txt 2>&1
done
echo "b200k nap=0" >> $OUT/ab_nap.txt; GS2D_NAP_WAIT=0 timeout -k 10 200 python scripts/dev/stage_ms.py 1 --workload b200k >> $OUT/ab_nap.txt 2>&1
echo "b200k nap=1" >> $OUT/ab_nap.txt; timeout -k 10 200 python scripts/dev/stage_ms.py 1 --workload b200k >> $OUT/ab_nap.txt 2>&1
echo "b200k ahead=0" >> $OUT/ab_nap.txt; GS2D_LAUNCH_AHEAD=0 timeout -k 10 200 python scripts/dev/stage_ms.py 1 --workload b200k >> $OUT/ab_nap.txt 2>&1
echo "tracking nap=0" >> $OUT/ab_nap.txt; GS2D_NAP_WAIT=0 timeout -k 10 200 python scripts/dev/stage_ms.py 1 --workload tracking >> $OUT/ab_nap.txt 2>&1
echo "tracking nap=1" >> $OUT/ab_nap.txt; timeout -k 10 200 python scripts/dev/stage_ms.py 1 --workload tracking >> $OUT/ab_nap.txt 2>&1
cat $OUT/ab_nap.txt
