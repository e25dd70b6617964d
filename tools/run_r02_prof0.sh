cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_r02_acc0
export AVA_HIP_LIB_TAG=lab
export AVA_BN_ACC=0
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r02_acc0 -o r02 --output-format csv -- python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --global-batch 0 --no-loader-path > gpurun_out/prof_r02_acc0/bench.json 2> gpurun_out/prof_r02_acc0/bench.err
export AVA_BN_ACC=1
mkdir -p gpurun_out/prof_r02_acc1
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r02_acc1 -o r02 --output-format csv -- python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --global-batch 0 --no-loader-path > gpurun_out/prof_r02_acc1/bench.json 2> gpurun_out/prof_r02_acc1/bench.err
