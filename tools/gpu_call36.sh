#!/bin/bash
mkdir -p gpurun_out
cd /root/repo
O=gpurun_out/r02_chain_anatomy_v3_forms.txt
: > $O
echo "=== anatomy x1" >> $O; timeout 150 python tools/profile_chain.py cfg3 1 2>&1 | grep -v amdgpu | cut -c1-600 >> $O
echo "=== anatomy x64 (two streams per layer item, two head replicas), per layer" >> $O; WN_PROFILE_LAYERS=1 timeout 150 python tools/profile_chain.py cfg3 64 2>&1 | grep -v amdgpu | cut -c1-600 >> $O
echo "=== anatomy x128" >> $O; timeout 150 python tools/profile_chain.py cfg3 128 2>&1 | grep -v amdgpu | cut -c1-600 >> $O
tail -5 $O
