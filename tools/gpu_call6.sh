#!/bin/bash
mkdir -p gpurun_out
cd /root/repo
out=gpurun_out/r02_v3_l0_prefetch.txt
: > $out
timeout 120 python tools/quick_check.py cfg3 7 >> $out 2>&1
timeout 120 python tools/quick_check.py cfg1 5 >> $out 2>&1
for ns in 16 32 48 64 96; do
  echo "=== v3 cfg3 x$ns" >> $out
  timeout 200 python tools/profile_chain.py cfg3 $ns 2>&1 | grep -v "amdgpu.ids\|^  layer [1-4]\|sampler [1-9]" >> $out
done
cat $out
