#!/bin/bash
# product forward without torch, two processes (NOTES item 44)
mkdir -p gpurun_out/twoproc
R=tools/lab/two_proc_fold
{
timeout 200 $R solo 500 8 0
timeout 400 $R lockstep 3000 8 0
timeout 400 $R mixed 3000 8 0
timeout 400 $R lockstep 1000 8 1
timeout 400 $R turns 1000 8 0
timeout 400 $R lockstep 600 64 0
} > gpurun_out/twoproc/fold.log 2>&1
grep -v amdgpu.ids gpurun_out/twoproc/fold.log | tail -60
