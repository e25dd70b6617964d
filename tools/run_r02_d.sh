cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -m gpu -q > gpurun_out/d_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/d_pytest.log
tail -30 gpurun_out/d_pytest.log
timeout 900 python bench.py --no-cpu-baseline --steps 50 --warmup 10 --global-batch 0 --no-loader-path > gpurun_out/d_bench128.json 2> gpurun_out/d_bench128.err; cat gpurun_out/d_bench128.json; tail -2 gpurun_out/d_bench128.err
timeout 900 python bench.py --no-cpu-baseline --steps 30 --warmup 10 --global-batch 0 --no-loader-path --height 256 --width 256 --z-dim 128 --per-gpu-batch 64 > gpurun_out/d_bench256.json 2> gpurun_out/d_bench256.err; cat gpurun_out/d_bench256.json; tail -2 gpurun_out/d_bench256.err
