cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -m gpu -q > gpurun_out/b_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/b_pytest.log
tail -25 gpurun_out/b_pytest.log
timeout 600 python tools/mmd_bench.py > gpurun_out/b_mmd.json 2> gpurun_out/b_mmd.err; cat gpurun_out/b_mmd.json; tail -3 gpurun_out/b_mmd.err
timeout 900 python bench.py --no-cpu-baseline --steps 50 --warmup 10 > gpurun_out/b_bench.json 2> gpurun_out/b_bench.err; cat gpurun_out/b_bench.json; tail -3 gpurun_out/b_bench.err
