#!/bin/bash
# round 3, GPU call 4: the skip group fetches the layer's input (critical group = pure compute)
mkdir -p gpurun_out
cd /root/repo
O=gpurun_out/r03_call4.txt
: > $O
for n in 1 2 7 64; do timeout 120 python tools/quick_check.py cfg3 $n 2>&1 | grep quick_check >> $O; done
for m in 0 3; do
  echo "##### WN_V3_MODE=$m" >> $O
  for n in 1 16 32 48 64 96 128; do WN_V3_MODE=$m timeout 120 python tools/rate.py cfg3 $n 2000 2 2>&1 | grep "samples/s" >> $O; done
done
echo "=== anatomy x64 mode 0" >> $O; WN_V3_MODE=0 timeout 150 python tools/profile_chain.py cfg3 64 2>&1 | grep -v amdgpu | cut -c1-500 >> $O
cat $O
