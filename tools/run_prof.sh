set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
python bench.py --steps 50 --warmup 10 > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err
rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o v20 --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_prof.json 2> gpurun_out/bench_prof.err
ls -R gpurun_out/prof | head -20
cat gpurun_out/bench_full.json
