cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
export AVA_HIP_LIB_TAG=lab
for v in 0 1 0 1; do
  export AVA_SKIP_BN_FIN=$v
  echo "== AVA_SKIP_BN_FIN=$v"; timeout 300 python bench.py --no-cpu-baseline --steps 100 --warmup 40 --global-batch 0 --no-loader-path --no-roofline --lr 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
