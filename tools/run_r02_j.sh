cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -m gpu -q -x > gpurun_out/j_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/j_pytest.log
tail -4 gpurun_out/j_pytest.log
export AVA_HIP_LIB_TAG=lab
for rep in 1 2; do
for v in 0 1; do
  export AVA_BN_ACC=$v
  echo -n "AVA_BN_ACC=$v  "; timeout 300 python bench.py --no-cpu-baseline --steps 150 --warmup 40 --global-batch 0 --no-loader-path --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done; done
