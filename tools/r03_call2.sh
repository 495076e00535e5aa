#!/bin/bash
# round 3, GPU call 2: batched head poll (product build) + sleep of the skip / queue groups after barrier B (variants)
mkdir -p gpurun_out
cd /root/repo
O=gpurun_out/r03_call2.txt
: > $O
for n in 1 7 64; do timeout 120 python tools/quick_check.py cfg3 $n 2>&1 | grep quick_check >> $O; done
for n in 1 32 64 96; do timeout 120 python tools/rate.py cfg3 $n 2000 2 2>&1 | grep "samples/s" >> $O; done
for v in S4 S8 Q4 Q8 S4Q4 S8Q8; do
  echo "##### variant $v" >> $O
  WN_DEV_LIB=tools/variants/libwn_$v.so timeout 120 python tools/quick_check.py cfg3 64 2>&1 | grep quick_check >> $O
  for n in 64 96; do WN_DEV_LIB=tools/variants/libwn_$v.so timeout 120 python tools/rate.py cfg3 $n 2000 2 2>&1 | grep "samples/s" >> $O; done
done
echo "=== anatomy x64" >> $O; timeout 150 python tools/profile_chain.py cfg3 64 2>&1 | grep -v amdgpu | grep "ring tail\|loop period\|hand-off x'\|layers>0" | cut -c1-700 >> $O
echo "=== anatomy x1" >> $O; timeout 150 python tools/profile_chain.py cfg3 1 2>&1 | grep -v amdgpu | cut -c1-400 >> $O
cat $O
