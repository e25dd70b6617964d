cd $GRAFT_REPO_ROOT
python3 bench.py --z-dim 64 --no-cpu-baseline > gpurun_out/final/bench_z64.json 2> gpurun_out/final/bench_z64.err
tail -3 gpurun_out/final/bench_z64.err; tail -c 900 gpurun_out/final/bench_z64.json
