#!/bin/bash
mkdir -p gpurun_out
cd /root/repo
out=gpurun_out/r02_v3_ablations.txt
: > $out
for v in abl1 abl2 abl3; do
  echo "##### $v (timing only, results wrong)" >> $out
  for ns in 32 64 96; do WN_DEV_LIB=tools/variants/libwn_$v.so timeout 200 python tools/rate.py cfg3 $ns 2000 1 2>&1 | grep -v amdgpu.ids >> $out; done
done
echo "=== abl3 anatomy x64" >> $out
WN_DEV_LIB=tools/variants/libwn_abl3.so timeout 200 python tools/profile_chain.py cfg3 64 2>&1 | grep -v "amdgpu.ids\|^  layer [1-4]\|sampler [1-9]" >> $out
cat $out
