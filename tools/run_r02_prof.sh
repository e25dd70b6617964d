# rocprofv3 kernel stats of the bench (20 steps) -> gpurun_out/prof_r02/
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_r02
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r02 -o r02 --output-format csv -- python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --global-batch 0 --no-loader-path > gpurun_out/prof_r02/bench.json 2> gpurun_out/prof_r02/bench.err
find gpurun_out/prof_r02 -name "*kernel_stats.csv" | head
