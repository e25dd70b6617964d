# round-2 GPU call A: full -m gpu suite, fp64 gradient table, default bench line, 2-rank gloo bench
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/a_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/a_pytest.log
tail -15 gpurun_out/a_pytest.log
timeout 900 python tools/grad_fp64.py > gpurun_out/a_grad_fp64.log 2>&1; tail -70 gpurun_out/a_grad_fp64.log
timeout 900 python bench.py > gpurun_out/a_bench.json 2> gpurun_out/a_bench.err; cat gpurun_out/a_bench.json; tail -3 gpurun_out/a_bench.err
