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
    rocprofv3 --kernel-trace --stats -d $out/prof -o bench --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loader-path --global-batch 0 > $out/bench_prof.json 2> $out/bench_prof.err
    find $out/prof -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
    rm -rf $out/prof
    ;;
  reserve)
    timeout 600 python -m pytest tests/test_gpu_reserve.py -x -q > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log
    timeout 600 python tools/reserve_bench.py > $out/reserve_bench.log 2>&1
    tail -n 5 $out/pytest.log; cat $out/reserve_bench.log
    ;;
  final)
    # the round's evidence in one call: kernel stats, HBM traffic (two --pmc passes), SQ counters, the bench lines
    cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
    R=$PWD
    rocprofv3 --kernel-trace --stats -d $out/prof -o bench --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loader-path --global-batch 0 > $out/bench_under_rocprof.json 2> $out/bench_under_rocprof.err
    find $out/prof -name "*kernel_stats.csv" -exec cp {} $out/final_kernel_stats.csv \;
    rm -rf $out/prof
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o f -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --global-batch 0 --no-loader-path --no-roofline > $out/pmc_fetch.log 2>&1
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o w -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --global-batch 0 --no-loader-path --no-roofline > $out/pmc_write.log 2>&1
    python3 tools/pmc_traffic.py $(find $out/pmc_fetch -name "*counter_collection.csv" | head -1) $(find $out/pmc_write -name "*counter_collection.csv" | head -1) > $out/pmc_traffic.json
    rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $out/pmc_sq -o sq -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --global-batch 0 --no-loader-path --no-roofline > $out/pmc_sq.log 2>&1
    python3 tools/pmc_sq.py $(find $out/pmc_sq -name "*counter_collection.csv" | head -1) > $out/pmc_sq.json
    rm -rf $out/pmc_fetch $out/pmc_write $out/pmc_sq
    mkdir -p profiles/r03 && cp $out/pmc_traffic.json profiles/r03/pmc_traffic.json      # so that the bench below quotes this build's traffic
    python3 bench.py > $out/bench_final.json 2> $out/bench_final.err
    python3 bench.py --z-dim 64 --no-cpu-baseline --no-loader-path > $out/bench_z64.json 2> $out/bench_z64.err
    python3 bench.py --per-gpu-batch 128 --global-batch 0 --no-cpu-baseline --no-loader-path > $out/bench_B128.json 2> $out/bench_B128.err
    python3 bench.py --height 256 --width 256 --z-dim 128 --per-gpu-batch 64 --global-batch 0 --steps 50 --no-cpu-baseline --no-loader-path > $out/bench_256x256_z128_B64_fp32.json 2> $out/bench_256_fp32.err
    python3 bench.py --height 256 --width 256 --z-dim 128 --per-gpu-batch 64 --global-batch 0 --steps 50 --dtype bf16 --no-cpu-baseline --no-loader-path > $out/bench_256x256_z128_B64_bf16act.json 2> $out/bench_256_bf16.err
    tail -c 400 $out/bench_final.json
    ;;
  pair)
    # the paired weight-gradient launch of the 16 x 16 layers: bit-identity against separate launches, tests, step time
    for B in 8 256; do
      AB_B=$B AVA_HIP_LIB_TAG=lab AVA_WGRAD_PAIR=0 timeout 300 python tools/ab_grads.py dump /tmp/a$B.npz > /dev/null 2>&1
      AB_B=$B AVA_HIP_LIB_TAG=lab AVA_WGRAD_PAIR=1 timeout 300 python tools/ab_grads.py dump /tmp/b$B.npz > /dev/null 2>&1
      AB_B=$B timeout 300 python tools/ab_grads.py dump /tmp/c$B.npz > /dev/null 2>&1
      echo "B=$B lab pair off vs on" >> $out/ab.log; python tools/ab_grads.py diff /tmp/a$B.npz /tmp/b$B.npz 2>&1 | head -n 4 >> $out/ab.log
      echo "B=$B lab pair off vs product" >> $out/ab.log; python tools/ab_grads.py diff /tmp/a$B.npz /tmp/c$B.npz 2>&1 | head -n 4 >> $out/ab.log
    done
    cat $out/ab.log
    timeout 1500 python -m pytest tests/test_gpu_step.py tests/test_gpu_graph.py tests/test_gpu_callers.py tests/test_gpu_autograd_semantics.py -x -q > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log
    timeout 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-loader-path > $out/bench.json 2> $out/bench.err
    AVA_HIP_LIB_TAG=lab AVA_WGRAD_PAIR=0 timeout 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-loader-path > $out/bench_lab_off.json 2> $out/bench_lab_off.err
    AVA_HIP_LIB_TAG=lab AVA_WGRAD_PAIR=1 timeout 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-loader-path > $out/bench_lab_on.json 2> $out/bench_lab_on.err
    grep -o '"ms_per_step": [0-9.]*' $out/bench.json $out/bench_lab_off.json $out/bench_lab_on.json
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
