#!/bin/bash
# Phase ablation of the wave-specialised fused backward kernels: builds
# csrc/libava_hip_abl<bits>.so with one phase compiled out (RESULTS ARE WRONG, timing only) and prints the step time per
# variant.  bits: 1 = per-element BatchNorm sums, 2 = dx stores, 4 = data-gradient MFMAs, 8 = weight-gradient phase,
# 16 = the staging waves' global loads.  Run on the GPU box:  bash tools/ablate_fused.sh "2 4 8 16 12"
set -e
cd "$(dirname "$0")/.."
C=autoencoded-vocal-analysis_amd/csrc
cp $C/conv_fused.hip /tmp/conv_fused.orig
patch -p1 -s < tools/lab/fused_ablation.patch
for b in ${1:-"1 2 4 8 16 12"}; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on -DAVA_ABL=$b -c $C/conv_fused.hip -o /tmp/cf_abl$b.o
  ( cd $C && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC conv_dispatch.o conv_mfma.o conv_ws.o /tmp/cf_abl$b.o conv_thin.o conv_thin_perop.o gemm.o bn.o misc.o mmd.o feed.o spec.o model.o -o libava_hip_abl$b.so -Wl,-rpath,/opt/rocm/lib -lpthread )
done
cp /tmp/conv_fused.orig $C/conv_fused.hip
for tag in "" $(for b in ${1:-"1 2 4 8 16 12"}; do echo abl$b; done) ""; do
  AVA_HIP_LIB_TAG=$tag python3 bench.py --no-cpu-baseline --no-loader-path --global-batch 0 --lr 0 --steps 60 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tag=%-6s' % '$tag', d['ms_per_step'], 'ms/step  conv backward', d['roofline']['ms_per_step_by_category']['conv_bwd_data'])"
done
rm -f $C/libava_hip_abl*.so
