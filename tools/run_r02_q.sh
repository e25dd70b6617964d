cd $GRAFT_REPO_ROOT
export AVA_HIP_LIB_TAG=lab
for v in 0 1; do
  if [ $v = 1 ]; then export AVA_FUSED16=1; fi
  python3 bench.py --no-cpu-baseline --no-loader-path --global-batch 0 > gpurun_out/q_bench_$v.json 2> gpurun_out/q_bench_$v.err
  python3 - <<PY
import json
d=json.loads([l for l in open("gpurun_out/q_bench_$v.json") if l.startswith("{")][-1])
print("FUSED16=$v", d["value"], d["ms_per_step"], d["roofline"]["ms_per_step_by_category"], d["roofline"]["launch_groups_per_step"])
PY
done
AVA_FUSED16=1 python3 -m pytest tests/test_gpu_step.py -x -q -m gpu -k "golden or noise_floor" 2>&1 | tail -3
