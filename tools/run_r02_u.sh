cd $GRAFT_REPO_ROOT
for tag in "" priof priow prio2 "" priof priow prio2; do
  AVA_HIP_LIB_TAG=$tag python3 bench.py --no-cpu-baseline --no-loader-path --global-batch 0 --no-roofline > gpurun_out/u_bench.json 2> gpurun_out/u_bench.err
  python3 - <<PY
import json
d=json.loads([l for l in open("gpurun_out/u_bench.json") if l.startswith("{")][-1])
print("tag='$tag'", d["value"], d["ms_per_step"])
PY
done
