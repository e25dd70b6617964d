cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_gpu_spec.py -x -q -m gpu > gpurun_out/k_spec.log 2>&1
tail -30 gpurun_out/k_spec.log
