cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_gpu_graph.py -x -q -m gpu 2>&1 | tail -15
