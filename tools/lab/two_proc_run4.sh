#!/bin/bash
mkdir -p gpurun_out/twoproc
{
for k in 9 10; do timeout 300 tools/lab/two_proc_repro streams 4 50 64 512 $k; done
echo "=== subject built with -DSAFE_SPLAT (no op_sel on the packed FMA's broadcast operand) ==="
for k in 8 4; do timeout 300 tools/lab/two_proc_repro_safe streams 4 50 64 512 $k; done
timeout 300 tools/lab/two_proc_repro_safe streams 4 50 256 512 8
} > gpurun_out/twoproc/standalone_streams_k9_safe.log 2>&1
grep -v amdgpu.ids gpurun_out/twoproc/standalone_streams_k9_safe.log | grep -v "vs ref" | tail -40
