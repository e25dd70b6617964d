#!/bin/bash
# one gpurun call: the stand-alone two-process reproducer in its three modes, then the product probe for comparison
mkdir -p gpurun_out/twoproc
R=tools/lab/two_proc_repro
{
timeout 300 $R solo 4 50 8 128
timeout 600 $R lockstep 40 50 8 128
timeout 600 $R lockstep 10 50 256 512
timeout 600 $R lockstep 10 50 64 512
timeout 300 $R turns 10 50 8 128
} > gpurun_out/twoproc/standalone.log 2>&1
tail -40 gpurun_out/twoproc/standalone.log
timeout 900 python tools/lab/share_probe.py 20 0 > gpurun_out/twoproc/share_probe.log 2>&1
tail -30 gpurun_out/twoproc/share_probe.log
