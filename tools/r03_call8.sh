#!/bin/bash
mkdir -p gpurun_out
cd /root/repo
O=gpurun_out/r03_call8.txt
: > $O
for n in 1 2 7 64; do timeout 120 python tools/quick_check.py cfg3 $n 2>&1 | grep quick_check >> $O; done
for n in 1 16 32 48 64 96 128; do timeout 120 python tools/rate.py cfg3 $n 2000 2 2>&1 | grep "samples/s" >> $O; done
echo "=== anatomy x64" >> $O; timeout 150 python tools/profile_chain.py cfg3 64 2>&1 | grep -v amdgpu | grep "loop period\|multi\|skip group\|hand-off\|ring tail\|layers>0\|head (\|sampler 0" | cut -c1-500 >> $O
echo "=== anatomy x1" >> $O; timeout 150 python tools/profile_chain.py cfg3 1 2>&1 | grep -v amdgpu | cut -c1-300 >> $O
( time timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 ) >> $O 2>&1
cat $O
