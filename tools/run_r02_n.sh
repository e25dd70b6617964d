cd $GRAFT_REPO_ROOT
python3 -m pytest tests -x -q -m gpu > gpurun_out/n_pytest.log 2>&1; tail -5 gpurun_out/n_pytest.log
python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/n_smoke.log 2>&1; tail -3 gpurun_out/n_smoke.log
python3 bench.py > gpurun_out/n_bench.json 2> gpurun_out/n_bench.err; tail -c 1500 gpurun_out/n_bench.json
