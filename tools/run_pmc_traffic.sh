# HBM traffic per kernel: two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of the bench command
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -o f -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --global-batch 0 --no-loader-path --no-roofline > $R/gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -o w -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --global-batch 0 --no-loader-path --no-roofline > $R/gpurun_out/pmc_write.log 2>&1
ls $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write
