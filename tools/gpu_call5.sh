#!/bin/bash
mkdir -p gpurun_out
cd /root/repo
out=gpurun_out/r02_v3_samplers.txt
: > $out
for smp in 4 8; do
  for ns in 32 64; do
    echo "=== v3 cfg3 x$ns WN_SAMPLERS=$smp" >> $out
    WN_SAMPLERS=$smp timeout 200 python tools/profile_chain.py cfg3 $ns 2>&1 | grep -v "amdgpu.ids\|^  layer [1-4]" >> $out
  done
done
echo "=== v2 two chains cfg3 x64 WN_SAMPLERS=8" >> $out
WN_KERNEL=v2 WN_SAMPLERS=8 timeout 200 python tools/profile_chain.py cfg3 64 2>&1 | grep -v "amdgpu.ids\|^  layer [1-4]" >> $out
cat $out
