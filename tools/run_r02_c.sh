cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -x -k "odd_batches or noise_floor or flip_free or conv_forward or feed or mmd" > gpurun_out/c_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/c_pytest.log
tail -8 gpurun_out/c_pytest.log
timeout 600 python tools/feed_bench.py > gpurun_out/c_feed.log 2>&1; cat gpurun_out/c_feed.log
export AVA_HIP_LIB_TAG=lab
for s in default 2 4 8 16 32; do
  if [ $s = default ]; then unset AVA_GEMM_SPLITS; else export AVA_GEMM_SPLITS=$s; fi
  echo "== AVA_GEMM_SPLITS=$s"; timeout 300 python tools/gemm_bench.py 2>&1 | grep -E "fc1|fc8"
done > gpurun_out/c_gemm.log 2>&1
cat gpurun_out/c_gemm.log
