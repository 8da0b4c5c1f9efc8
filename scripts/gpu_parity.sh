set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m gaus_slam_amd.build 2>&1 | tail -2
make -s -C oracle
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -40
