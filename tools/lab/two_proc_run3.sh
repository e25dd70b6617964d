#!/bin/bash
# which neighbour disturbs convt7's forward, and is a second process needed? (NOTES item 44)
mkdir -p gpurun_out/twoproc
R=tools/lab/two_proc_fold
{
for k in gemm conv thin adam; do timeout 300 $R pair 3000 8 0 $k; done
for k in gemm conv thin adam; do timeout 300 $R streams 3000 8 0 $k; done
timeout 300 $R pair 600 64 0 gemm
timeout 300 $R streams 600 64 0 gemm
} > gpurun_out/twoproc/neighbours.log 2>&1
grep -v amdgpu.ids gpurun_out/twoproc/neighbours.log | grep -v "partial rows differ" | tail -60
