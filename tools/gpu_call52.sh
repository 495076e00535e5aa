#!/bin/bash
mkdir -p gpurun_out
cd /root/repo
O=gpurun_out/r02_v3_split_gate.txt
: > $O
for ns in 1 7 48; do timeout 120 python tools/quick_check.py cfg3 $ns 2>&1 | grep -v amdgpu >> $O; done
for ns in 1 16 32 48 64 64 96; do timeout 120 python tools/rate.py cfg3 $ns 2>&1 | grep -v amdgpu >> $O; done
cat $O
