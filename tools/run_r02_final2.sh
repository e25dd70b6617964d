# final-build evidence, second pass (after the GEMM plan change and f4): same passes as run_r02_final.sh + the other configs
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final; bash tools/run_r02_final.sh > gpurun_out/final/all.log 2>&1
cd $GRAFT_REPO_ROOT
python3 bench.py --z-dim 64 --no-cpu-baseline > gpurun_out/final/bench_z64.json 2> gpurun_out/final/bench_z64.err
python3 bench.py --height 256 --width 256 --z-dim 128 --batch 64 --dtype bf16 --no-cpu-baseline --no-loader-path --global-batch 0 > gpurun_out/final/bench_256_bf16.json 2> gpurun_out/final/bench_256_bf16.err
python3 bench.py --height 256 --width 256 --z-dim 128 --batch 64 --no-cpu-baseline --no-loader-path --global-batch 0 > gpurun_out/final/bench_256_f32.json 2> gpurun_out/final/bench_256_f32.err
python3 bench.py --batch 128 --no-cpu-baseline --no-loader-path --global-batch 0 > gpurun_out/final/bench_B128.json 2> gpurun_out/final/bench_B128.err
for f in bench_default bench_z64 bench_256_bf16 bench_256_f32 bench_B128; do python3 - <<PY
import json
d=json.loads([l for l in open("gpurun_out/final/$f.json") if l.startswith("{")][-1])
print("$f", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["conv_family_ms_per_step"], d.get("shotgun_path",{}).get("train_epoch_fed_by_device_spectrograms"))
PY
done
