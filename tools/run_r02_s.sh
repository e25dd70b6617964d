cd $GRAFT_REPO_ROOT
python3 tools/callers_noise.py 2>&1 | tail -2
python3 -m pytest tests/test_gpu_callers.py tests/test_gpu_kernels.py -x -q -m gpu 2>&1 | tail -3
