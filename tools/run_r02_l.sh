cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python3 tools/spec_bench.py > gpurun_out/l_spec_bench.json 2> gpurun_out/l_spec_bench.err
cat gpurun_out/l_spec_bench.json; tail -3 gpurun_out/l_spec_bench.err
mkdir -p gpurun_out/l_prof
rocprofv3 --kernel-trace --stats -d gpurun_out/l_prof -o spec --output-format csv -- python3 tools/spec_bench.py 256 10 > /dev/null 2>&1
grep -E "spec_|Name" $(find gpurun_out/l_prof -name "*kernel_stats.csv" | head -1) | head
cp $(find gpurun_out/l_prof -name "*kernel_stats.csv" | head -1) gpurun_out/l_spec_kernel_stats.csv
rm -rf gpurun_out/l_prof
