set -e
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04m
mkdir -p $OUT
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1 || { tail -60 $OUT/pytest_gpu.log; exit 1; }
tail -3 $OUT/pytest_gpu.log
timeout -k 10 400 python bench.py --no-cpu-baseline --json-out $OUT/bench.json > $OUT/bench.log 2>&1 || true
python3 -c "import json;d=json.load(open('$OUT/bench.json'));print(d['value'],d['ms_per_step'],d['config']['cold_value'],d['config']['reference_binning_value'],d['config']['host_cpu_fraction'],d['roofline']['stage_ms'])"
