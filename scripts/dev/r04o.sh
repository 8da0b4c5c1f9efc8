set -e
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04o
mkdir -p $OUT
rm -f $OUT/ab.txt
for w in op b200k scannetpp tracking; do
  WL=""; [ $w != op ] && WL="--workload $w"
  for nap in 0 1; do
    GS2D_NAP_WAIT=$nap timeout -k 10 300 python bench.py $WL --steps 40 --warmup 5 --no-cpu-baseline --no-extra-legs --json-out $OUT/b_${w}_$nap.json > /dev/null 2>&1
    python3 -c "import json;d=json.load(open('$OUT/b_${w}_$nap.json'));print('$w nap=$nap', d['value'], d['ms_per_step'], 'host_cpu', d['config']['host_cpu_fraction'])" >> $OUT/ab.txt
  done
done
cat $OUT/ab.txt
