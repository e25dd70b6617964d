cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "gemm" 2>&1 | tail -2
python3 tools/gemm_bench.py 2>&1 | grep -E "fc1|fc8|\{"
python3 tools/gemm_peak.py 2>&1 | tail -8
