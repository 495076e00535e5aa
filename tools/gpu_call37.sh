#!/bin/bash
mkdir -p gpurun_out
cd /root/repo
O=gpurun_out/r02_v3_head_placement.txt
: > $O
echo "##### WN_HEAD_XCD=last: head + samplers in the XCD of the last layers" >> $O
export WN_HEAD_XCD=last
timeout 120 python tools/quick_check.py cfg3 7 2>&1 | grep -v amdgpu >> $O
for ns in 1 32 64 96; do timeout 120 python tools/rate.py cfg3 $ns 2>&1 | grep -v amdgpu >> $O; done
timeout 120 python tools/profile_chain.py cfg3 64 2>&1 | grep -v amdgpu | grep "head \|sampler 0\|hand-off\|layers>0\|loop period\|layer  0\|layer 49" >> $O
cat $O
