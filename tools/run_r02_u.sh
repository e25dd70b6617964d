cd $GRAFT_REPO_ROOT
for tag in "" abl2 abl4 abl8 abl16 abl12 ""; do
  AVA_HIP_LIB_TAG=$tag python3 bench.py --no-cpu-baseline --no-loader-path --global-batch 0 --lr 0 --steps 60 > gpurun_out/u_bench.json 2> gpurun_out/u_bench.err
  python3 - <<PY
import json
try:
    d=json.loads([l for l in open("gpurun_out/u_bench.json") if l.startswith("{")][-1])
    print("tag='$tag'", d["value"], d["ms_per_step"], d["roofline"]["ms_per_step_by_category"]["conv_bwd_data"])
except Exception as e:
    print("tag='$tag' failed", e); print(open("gpurun_out/u_bench.err").read()[-400:])
PY
done
