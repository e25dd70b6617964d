cd $GRAFT_REPO_ROOT
for tag in "" abl "" abl; do
  AVA_HIP_LIB_TAG=$tag python3 bench.py --no-cpu-baseline --no-loader-path --global-batch 0 --lr 0 > gpurun_out/u_bench.json 2> gpurun_out/u_bench.err
  python3 - <<PY
import json
d=json.loads([l for l in open("gpurun_out/u_bench.json") if l.startswith("{")][-1])
print("tag='$tag'", d["value"], d["ms_per_step"], d["roofline"]["ms_per_step_by_category"]["conv_bwd_data"])
PY
done
