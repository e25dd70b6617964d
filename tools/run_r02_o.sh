cd $GRAFT_REPO_ROOT
export AVA_HIP_LIB_TAG=lab
echo "== BM=64 (all M=256 shapes)"; AVA_GEMM_BM=64 python3 tools/gemm_bench.py 2>&1 | grep -E "fc1|fc8"
echo "== BM=64 splits 1"; AVA_GEMM_BM=64 AVA_GEMM_SPLITS=1 python3 tools/gemm_bench.py 2>&1 | grep -E "fc1 +dX|fc8 +fwd"
echo "== BM=64 splits 2"; AVA_GEMM_BM=64 AVA_GEMM_SPLITS=2 python3 tools/gemm_bench.py 2>&1 | grep -E "fc1 +dX|fc8 +fwd"
