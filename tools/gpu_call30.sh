#!/bin/bash
mkdir -p gpurun_out
cd /root/repo
O=gpurun_out/r02_v3_interleaved_anatomy.txt
: > $O
echo "##### WN_V3_MODE=3 anatomy x64" >> $O; WN_V3_MODE=3 timeout 150 python tools/profile_chain.py cfg3 64 2>&1 | grep -v amdgpu | cut -c1-420 >> $O
echo "##### WN_V3_MODE=0 anatomy x16" >> $O; WN_V3_MODE=0 timeout 150 python tools/profile_chain.py cfg3 16 2>&1 | grep -v amdgpu | cut -c1-420 | grep -v "^  layer\|hand-off by\|^ 0" >> $O
cat $O
