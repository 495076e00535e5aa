#!/bin/bash
mkdir -p gpurun_out
cd /root/repo
out=gpurun_out/r02_v3_handoff.txt
: > $out
for ns in 16 32 48 64; do
echo "=== v3 anatomy x$ns" >> $out
timeout 200 python tools/profile_chain.py cfg3 $ns 2>&1 | grep -v "amdgpu.ids\|^  layer [1-4]\|sampler [1-9]" >> $out
done
cat $out
