#!/bin/bash
# phase ablation of the limb GEMM on the six fc1 / fc8 products (lab build, AVA_GEMM_LIMB_DBG bits: 1 no MFMA, 2 no split / LDS write,
# 4 no loads, 8 no epilogue stores); timing only
export AVA_HIP_LIB_TAG=lab
for d in 0 1 2 4 8 3 6 7 15; do
  echo "== dbg $d"; AVA_GEMM_LIMB_DBG=$d python3 tools/gemm_bench.py 256 2>/dev/null | grep -E "^fc1|^fc8" | cut -c1-75
done
