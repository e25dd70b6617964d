#!/bin/bash
# Round-6 GPU passes, one gpurun call each:  tools/run_r06.sh <pass> [tag]   (output under gpurun_out/r06_<pass><tag>/)
set -u
pass=${1:-kern}
tag=${2:-}
out=gpurun_out/r06_$pass$tag
mkdir -p $out
prof() {   # prof <name>: kernel trace of a short bench -> $out/<name>_kernel_stats.csv
  cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
  timeout 400 rocprofv3 --kernel-trace --stats -d $out/prof -o bench --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loader-path --global-batch 0 > $out/$1_bench_prof.json 2> $out/$1_bench_prof.err
  find $out/prof -name "*kernel_stats.csv" -exec cp {} $out/$1_kernel_stats.csv \;
  rm -rf $out/prof
}
case $pass in
  kern)
    # per-layer kernel parity, whole-step gradient parity, then the step under the kernel trace
    timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "backward_data_and_wgrad" > $out/pytest_kern.log 2>&1; echo "pytest rc $?" >> $out/pytest_kern.log
    tail -n 6 $out/pytest_kern.log
    timeout 1500 python -m pytest tests/test_gpu_step.py -x -q -k "flip or fp64_noise or odd_batches or golden or full_batch" > $out/pytest_step.log 2>&1; echo "pytest rc $?" >> $out/pytest_step.log
    tail -n 6 $out/pytest_step.log
    prof step
    timeout 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-loader-path --global-batch 0 > $out/bench.json 2> $out/bench.err
    python3 tools/kstats.py $out/step_kernel_stats.csv 25 | head -40
    head -c 300 $out/bench.json
    ;;
  step)
    timeout 2000 python -m pytest tests/test_gpu_step.py tests/test_gpu_autograd_semantics.py tests/test_gpu_callers.py tests/test_gpu_graph.py -x -q > $out/pytest_step.log 2>&1; echo "pytest rc $?" >> $out/pytest_step.log
    tail -n 8 $out/pytest_step.log
    prof step
    timeout 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-loader-path --global-batch 0 > $out/bench.json 2> $out/bench.err
    python3 tools/kstats.py $out/step_kernel_stats.csv 65 | grep -E "fc_mid|skinny|latent|total"
    head -c 300 $out/bench.json
    ;;
  full)
    timeout 2400 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log
    tail -n 15 $out/pytest.log
    timeout 600 python bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err
    head -c 300 $out/bench.json
    ;;
  variants)
    # same-box A/B of library variants (tools/lab/build_variant.sh): kernel parity of each, then the step under the kernel trace
    for v in "" v1 v2 v3 v4 v5; do
      if [ -n "$v" ] && [ ! -f autoencoded-vocal-analysis_amd/csrc/libava_hip_$v.so ]; then continue; fi
      export AVA_HIP_LIB_TAG=$v
      timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "backward_data_and_wgrad" > $out/pytest_kern_$v.log 2>&1; echo "pytest rc $?" >> $out/pytest_kern_$v.log
      tail -n 3 $out/pytest_kern_$v.log
      prof step_$v
      echo "== variant '$v'"; python3 tools/kstats.py $out/step_${v}_kernel_stats.csv 65 | grep -E "bwd_fused|total"
    done
    ;;
  final)
    # the round's evidence in one call: kernel stats, HBM traffic (two --pmc passes), SQ counters, the bench lines
    cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
    timeout 400 rocprofv3 --kernel-trace --stats -d $out/prof -o bench --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loader-path --global-batch 0 > $out/bench_under_rocprof.json 2> $out/bench_under_rocprof.err
    find $out/prof -name "*kernel_stats.csv" -exec cp {} $out/final_kernel_stats.csv \;
    rm -rf $out/prof
    timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o f -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --global-batch 0 --no-loader-path --no-roofline > $out/pmc_fetch.log 2>&1
    timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o w -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --global-batch 0 --no-loader-path --no-roofline > $out/pmc_write.log 2>&1
    python3 tools/pmc_traffic.py $(find $out/pmc_fetch -name "*counter_collection.csv" | head -1) $(find $out/pmc_write -name "*counter_collection.csv" | head -1) > $out/pmc_traffic.json
    timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $out/pmc_sq -o sq -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --global-batch 0 --no-loader-path --no-roofline > $out/pmc_sq.log 2>&1
    python3 tools/pmc_sq.py $(find $out/pmc_sq -name "*counter_collection.csv" | head -1) > $out/pmc_sq.json
    rm -rf $out/pmc_fetch $out/pmc_write $out/pmc_sq
    mkdir -p profiles/r06 && cp $out/pmc_traffic.json profiles/r06/pmc_traffic.json      # so that the bench below quotes this build's traffic
    timeout 900 python3 bench.py > $out/bench_final.json 2> $out/bench_final.err
    timeout 400 python3 bench.py --z-dim 64 --no-cpu-baseline --no-loader-path > $out/bench_z64.json 2> $out/bench_z64.err
    timeout 400 python3 bench.py --per-gpu-batch 128 --global-batch 0 --no-cpu-baseline --no-loader-path > $out/bench_B128.json 2> $out/bench_B128.err
    timeout 400 python3 bench.py --height 256 --width 256 --z-dim 128 --per-gpu-batch 64 --global-batch 0 --steps 50 --no-cpu-baseline --no-loader-path > $out/bench_256x256_z128_B64_fp32.json 2> $out/bench_256_fp32.err
    timeout 400 python3 bench.py --height 256 --width 256 --z-dim 128 --per-gpu-batch 64 --global-batch 0 --steps 50 --dtype bf16 --no-cpu-baseline --no-loader-path > $out/bench_256x256_z128_B64_bf16.json 2> $out/bench_256_bf16.err
    timeout 400 python3 bench.py --dtype bf16 --global-batch 0 --no-cpu-baseline --no-loader-path > $out/bench_128x128_bf16.json 2> $out/bench_128_bf16.err
    timeout 400 python3 bench.py --per-gpu-batch 1024 --global-batch 0 --steps 30 --no-cpu-baseline --no-loader-path > $out/bench_B1024.json 2> $out/bench_B1024.err
    tail -c 600 $out/bench_final.json
    ;;
  scratchenv)
    # does the HSA runtime's scratch handling explain why a kernel's time grows with its scratch size?  
    for e in none HSA_NO_SCRATCH_RECLAIM=1 HSA_NO_SCRATCH_THREAD_LIMITER=1 HSA_ENABLE_SCRATCH_ASYNC_RECLAIM=0 HSA_ENABLE_SCRATCH_ALT=1; do
      n=${e%%=*}
      if [ $e != none ]; then export $e; fi
      prof env_$n
      if [ $e != none ]; then unset $n; fi
      echo "== $e"; python3 tools/kstats.py $out/env_${n}_kernel_stats.csv 65 | grep -E "bwd_fused_limb|total" | cut -c1-140
    done
    ;;
  prof)
    prof step
    python3 tools/kstats.py $out/step_kernel_stats.csv 25 | head -60
    ;;
esac
