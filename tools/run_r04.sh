#!/bin/bash
# Round-4 GPU passes, one gpurun call each:  tools/run_r04.sh <pass> [tag]   (output under gpurun_out/r04_<pass><tag>/)
set -u
pass=${1:-kern}
tag=${2:-}
out=gpurun_out/r04_$pass$tag
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
  prof)
    prof step
    python3 tools/kstats.py $out/step_kernel_stats.csv 25 | head -60
    ;;
esac
