# final-build evidence of round 2: kernel stats, HBM traffic (two --pmc passes), SQ counters, the default bench line
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
bash tools/run_r02_prof.sh > gpurun_out/final/prof.log 2>&1
cp $(find gpurun_out/prof_r02 -name "*kernel_stats.csv" | head -1) gpurun_out/final/kernel_stats.csv
cp gpurun_out/prof_r02/bench.json gpurun_out/final/bench_under_rocprof.json
bash tools/run_pmc_traffic.sh > gpurun_out/final/pmc_traffic.log 2>&1
cd $R
python3 tools/pmc_traffic.py $(find gpurun_out/pmc_fetch -name "*counter_collection.csv" | head -1) $(find gpurun_out/pmc_write -name "*counter_collection.csv" | head -1) > gpurun_out/final/pmc_traffic.json
bash tools/run_pmc_sq.sh > gpurun_out/final/pmc_sq.log 2>&1
cd $R
python3 tools/pmc_sq.py $(find gpurun_out/pmc_sq -name "*counter_collection.csv" | head -1) > gpurun_out/final/pmc_sq.json
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_sq gpurun_out/prof_r02
mkdir -p profiles/r02 && cp gpurun_out/final/pmc_traffic.json profiles/r02/pmc_traffic.json   # so that the bench quotes this build's traffic
python3 bench.py > gpurun_out/final/bench_default.json 2> gpurun_out/final/bench_default.err
tail -c 600 gpurun_out/final/bench_default.json
