cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -m gpu -q -x > gpurun_out/f_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/f_pytest.log
tail -30 gpurun_out/f_pytest.log
timeout 900 python bench.py --no-cpu-baseline --steps 50 --warmup 10 --global-batch 0 --no-loader-path > gpurun_out/f_bench128.json 2> gpurun_out/f_bench128.err; cat gpurun_out/f_bench128.json; tail -2 gpurun_out/f_bench128.err
for dt in f32 bf16; do
timeout 900 python bench.py --no-cpu-baseline --steps 30 --warmup 10 --global-batch 0 --no-loader-path --height 256 --width 256 --z-dim 128 --per-gpu-batch 64 --dtype $dt > gpurun_out/f_bench256_$dt.json 2> gpurun_out/f_bench256_$dt.err; cat gpurun_out/f_bench256_$dt.json; tail -2 gpurun_out/f_bench256_$dt.err
done
timeout 900 python bench.py --no-cpu-baseline --steps 50 --warmup 10 --global-batch 0 --no-loader-path --dtype bf16 > gpurun_out/f_bench128_bf16.json 2> gpurun_out/f_bench128_bf16.err; cat gpurun_out/f_bench128_bf16.json
