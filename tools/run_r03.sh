#!/bin/bash
# Round-3 GPU passes, one gpurun call each:  tools/run_r03.sh <pass>   (output under gpurun_out/r03_<pass>/)
set -u
pass=${1:-gemm}
out=gpurun_out/r03_$pass
mkdir -p $out
case $pass in
  gemm)
    timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k gemm > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log
    timeout 300 python tools/gemm_bench.py > $out/gemm_bench.log 2>&1
    for cfg in 64_1 64_2 64_4 64_8 64_16 64_32 128_16; do
      bn=${cfg%_*}; sp=${cfg#*_}
      AVA_HIP_LIB_TAG=lab AVA_GEMM_LIMB_BN=$bn AVA_GEMM_LIMB_SPLITS=$sp timeout 300 python tools/gemm_bench.py 2>&1 | grep -E "fc1|fc8" > $out/gemm_bench_bn${bn}_s${sp}.log
    done
    AVA_HIP_LIB_TAG=lab AVA_GEMM_LIMB=0 timeout 300 python tools/gemm_bench.py 2>&1 | grep -E "fc1|fc8" > $out/gemm_bench_fp32.log
    timeout 600 python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err
    ;;
  step)
    timeout 1500 python -m pytest tests/test_gpu_step.py tests/test_gpu_kernels.py tests/test_gpu_autograd_semantics.py -x -q > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log
    timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench.json 2> $out/bench.err
    tail -n 15 $out/pytest.log
    ;;
  full)
    timeout 2400 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log
    timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench.json 2> $out/bench.err
    tail -n 15 $out/pytest.log
    ;;
  prof)
    cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
    rocprofv3 --kernel-trace --stats -d $out/prof -o bench --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_prof.json 2> $out/bench_prof.err
    find $out/prof -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
    rm -rf $out/prof
    ;;
  reserve)
    timeout 600 python -m pytest tests/test_gpu_reserve.py -x -q > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log
    timeout 600 python tools/reserve_bench.py > $out/reserve_bench.log 2>&1
    tail -n 5 $out/pytest.log; cat $out/reserve_bench.log
    ;;
  gemmabl)
    timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k gemm > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log
    for d in 0 1 2 4 8 3 6 7 15; do
      AVA_HIP_LIB_TAG=lab AVA_GEMM_LIMB_DBG=$d timeout 300 python tools/gemm_bench.py 2>&1 | grep -E "fc1|fc8" | cut -c1-75 > $out/abl_$d.log
    done
    ;;
  gemmprof)
    cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
    rocprofv3 --kernel-trace --stats -d $out/prof -o gemm --output-format csv -- python3 tools/gemm_bench.py > $out/gemm_bench.log 2>&1
    find $out/prof -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
    rm -rf $out/prof
    ;;
  *) echo "unknown pass $pass"; exit 2;;
esac
tail -n 5 $out/pytest.log
