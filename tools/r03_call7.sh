#!/bin/bash
mkdir -p gpurun_out
cd /root/repo
O=gpurun_out/r03_call7.txt
: > $O
for m in 0 3 4; do for n in 1 2 7 64; do WN_V3_MODE=$m timeout 120 python tools/quick_check.py cfg3 $n 2>&1 | grep quick_check | sed "s/^/mode $m: /" >> $O; done; done
for n in 1 16 32 48 64 96 128; do timeout 120 python tools/rate.py cfg3 $n 2000 2 2>&1 | grep "samples/s" >> $O; done
for m in 0 4; do for n in 1 16 32 40; do WN_V3_MODE=$m timeout 120 python tools/rate.py cfg3 $n 2000 2 2>&1 | grep "samples/s" | sed "s/^/mode $m: /" >> $O; done; done
echo "=== anatomy x64" >> $O; timeout 150 python tools/profile_chain.py cfg3 64 2>&1 | grep -v amdgpu | grep "loop period\|multi\|skip group\|hand-off x\|ring tail\|layers>0\|head (" | cut -c1-500 >> $O
echo "=== anatomy x1" >> $O; timeout 150 python tools/profile_chain.py cfg3 1 2>&1 | grep -v amdgpu | cut -c1-300 >> $O
cat $O
