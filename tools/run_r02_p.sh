cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_step.py -x -q -m gpu -k "gemm or golden or noise_floor or determin or full_batch" > gpurun_out/p_pytest.log 2>&1; tail -4 gpurun_out/p_pytest.log
python3 tools/gemm_bench.py 2>&1 | grep -E "fc1|fc8"
python3 bench.py --no-cpu-baseline --no-loader-path --global-batch 0 > gpurun_out/p_bench.json 2> gpurun_out/p_bench.err; python3 - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/p_bench.json") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["roofline"]["ms_per_step_by_category"])
PY
