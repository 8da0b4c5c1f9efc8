# Dev: per-kernel times of the binning chain for the product library and one variant (rocprofv3 --kernel-trace).
# usage (GPU box): bash scripts/dev/binexp2.sh <variant name> [workload=op] -> gpurun_out/binexp2/summary.txt
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
V=$1; W=${2:-op}; WL=""; [ "$W" != "op" ] && WL="--workload $W"
mkdir -p gpurun_out/binexp2
for v in product $V product $V; do
  if [ $v = product ]; then unset GS2D_LIB_PATH; else export GS2D_LIB_PATH=$GRAFT_REPO_ROOT/scripts/dev/variants/lib$v.so; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/binexp2/$v -o t -- python3 bench.py $WL --steps 40 --warmup 5 --no-cpu-baseline --no-extra-legs --prewarm-steps 200 > gpurun_out/binexp2/$v.log 2>&1 || echo "run $v failed"
  echo "== $v $W" >> gpurun_out/binexp2/summary.txt
  python3 scripts/dev/kernel_times.py gpurun_out/binexp2/$v 0.5 | grep -E "bin_|duplicate|preprocess|blend|window" >> gpurun_out/binexp2/summary.txt
  rm -rf gpurun_out/binexp2/$v
done
cat gpurun_out/binexp2/summary.txt
