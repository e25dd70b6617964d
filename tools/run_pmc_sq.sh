# wave-state counters per kernel (one rocprofv3 --pmc pass, kernel-trace only); summaries land in gpurun_out/pmc_sq
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $R/gpurun_out/pmc_sq -o sq -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --global-batch 0 --no-loader-path --no-roofline > $R/gpurun_out/pmc_sq.log 2>&1
ls -la $R/gpurun_out/pmc_sq; tail -2 $R/gpurun_out/pmc_sq.log
