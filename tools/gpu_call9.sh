#!/bin/bash
mkdir -p gpurun_out
cd /root/repo
out=gpurun_out/r02_v3_xpairs.txt
: > $out
timeout 120 python tools/quick_check.py cfg3 7 >> $out 2>&1
for ns in 32 48 64 96; do timeout 200 python tools/rate.py cfg3 $ns 2000 2 2>&1 | grep -v amdgpu.ids >> $out; done
cat $out
