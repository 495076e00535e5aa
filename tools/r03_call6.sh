#!/bin/bash
mkdir -p gpurun_out
cd /root/repo
O=gpurun_out/r03_call6.txt
: > $O
for n in 1 2 7 64; do timeout 120 python tools/quick_check.py cfg3 $n 2>&1 | grep quick_check >> $O; done
WN_V3_MODE=0 timeout 120 python tools/quick_check.py cfg3 64 2>&1 | grep quick_check >> $O
for v in product defer0 defer2; do
  echo "##### $v" >> $O
  lib=tools/variants/libwn_$v.so; [ $v = product ] && lib=pytorch-wavenet_amd/mi355_wavenet/libwn_mi355.so
  for m in 0 3; do for n in 1 16 32 48 64 96 128; do [ $m = 3 ] && [ $n -lt 48 ] && continue; WN_V3_MODE=$m WN_DEV_LIB=$lib timeout 120 python tools/rate.py cfg3 $n 2000 2 2>&1 | grep "samples/s" | sed "s/^/mode $m: /" >> $O; done; done
done
echo "=== anatomy x64 mode 0 (product build)" >> $O; WN_V3_MODE=0 timeout 150 python tools/profile_chain.py cfg3 64 2>&1 | grep -v amdgpu | grep "loop period\|multi\|skip group\|hand-off x\|ring tail\|layers>0\|layer 25" | cut -c1-500 >> $O
cat $O
