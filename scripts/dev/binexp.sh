set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/binexp
for v in product bin16; do
  if [ $v = product ]; then unset GS2D_LIB_PATH; else export GS2D_LIB_PATH=$GRAFT_REPO_ROOT/scripts/dev/variants/lib$v.so; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/binexp/$v -o t -- python3 bench.py --workload scannetpp --steps 20 --warmup 3 --no-cpu-baseline --no-extra-legs --prewarm-steps 0 > gpurun_out/binexp/$v.log 2>&1 || echo "run $v failed"
  echo "== $v" >> gpurun_out/binexp/summary.txt
  python3 scripts/dev/kernel_times.py gpurun_out/binexp/$v 0.5 | grep -E "bin_|duplicate|window" >> gpurun_out/binexp/summary.txt
  rm -rf gpurun_out/binexp/$v
done
cat gpurun_out/binexp/summary.txt
unset GS2D_LIB_PATH
for it in 4096 8192; do
  export GS2D_BIN_ITEMS_FORCE=$it
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/binexp/items$it -o t -- python3 bench.py --workload scannetpp --steps 20 --warmup 3 --no-cpu-baseline --no-extra-legs --prewarm-steps 0 > gpurun_out/binexp/items$it.log 2>&1 || echo "run failed"
  echo "== items $it" >> gpurun_out/binexp/summary.txt
  python3 scripts/dev/kernel_times.py gpurun_out/binexp/items$it 0.5 | grep -E "bin_|duplicate|window" >> gpurun_out/binexp/summary.txt
  rm -rf gpurun_out/binexp/items$it
done
cat gpurun_out/binexp/summary.txt
