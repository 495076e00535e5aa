#!/bin/bash
# round 3, GPU call 1: baseline rates, the timing ablations re-run in the two-streams-per-item form, ring-tail anatomy
mkdir -p gpurun_out
cd /root/repo
O=gpurun_out/r03_call1.txt
: > $O
for n in 1 64 96; do timeout 120 python tools/rate.py cfg3 $n 2000 2 2>&1 | grep "samples/s" >> $O; done
for a in 1 2 3; do
  echo "##### WN_V3_ABL=$a (timing only, results wrong; two streams per item at 64): 1 no skip-lane work, 2 no queue work, 3 neither" >> $O
  for n in 64; do WN_DEV_LIB=tools/variants/libwn_abl$a.so timeout 120 python tools/rate.py cfg3 $n 2000 2 2>&1 | grep "samples/s" >> $O; done
done
echo "=== anatomy x64" >> $O; timeout 150 python tools/profile_chain.py cfg3 64 2>&1 | grep -v amdgpu | cut -c1-700 >> $O
echo "=== anatomy x32 (one stream per item)" >> $O; timeout 150 python tools/profile_chain.py cfg3 32 2>&1 | grep -v amdgpu | cut -c1-700 >> $O
echo "=== anatomy x64 ABL=1" >> $O; WN_DEV_LIB=tools/variants/libwn_abl1.so timeout 150 python tools/profile_chain.py cfg3 64 2>&1 | grep -v amdgpu | cut -c1-700 >> $O
cat $O
