cd $GRAFT_REPO_ROOT
python3 -m pytest tests -x -q -m gpu > gpurun_out/t_pytest.log 2>&1; tail -3 gpurun_out/t_pytest.log
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
