cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python bench.py --cpu-protocol full > gpurun_out/g_bench_full.json 2> gpurun_out/g_bench_full.err; cat gpurun_out/g_bench_full.json
timeout 600 python bench.py --z-dim 64 --no-cpu-baseline --no-loader-path > gpurun_out/g_bench_z64.json 2> gpurun_out/g_bench_z64.err; cat gpurun_out/g_bench_z64.json
timeout 600 python bench.py --per-gpu-batch 128 --no-cpu-baseline --no-loader-path --global-batch 0 > gpurun_out/g_bench_B128.json 2> gpurun_out/g_bench_B128.err; cat gpurun_out/g_bench_B128.json
timeout 600 python bench.py --gpus 2 --backend gloo --no-cpu-baseline --steps 30 --warmup 10 > gpurun_out/g_bench_gloo2.json 2> gpurun_out/g_bench_gloo2.err; cat gpurun_out/g_bench_gloo2.json
