cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_gpu_step.py tests/test_gpu_callers.py -x -q -m gpu 2>&1 | tail -3
export AVA_HIP_LIB_TAG=lab
for v in 0 1 0 1; do
  AVA_BN_ACC=$v python3 bench.py --no-cpu-baseline --no-loader-path --global-batch 0 --no-roofline > gpurun_out/q_bench_$v.json 2> gpurun_out/q_bench_$v.err
  python3 - <<PY
import json
d=json.loads([l for l in open("gpurun_out/q_bench_$v.json") if l.startswith("{")][-1])
print("ACC=$v", d["value"], d["ms_per_step"])
PY
done
python3 bench.py --no-cpu-baseline --no-loader-path --global-batch 0 > gpurun_out/q_bench_p.json 2>/dev/null; python3 - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/q_bench_p.json") if l.startswith("{")][-1])
print("product", d["value"], d["ms_per_step"], d["roofline"]["launch_groups_per_step"], d["roofline"]["ms_per_step_by_category"])
PY
