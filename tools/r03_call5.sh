#!/bin/bash
mkdir -p gpurun_out
cd /root/repo
O=gpurun_out/r03_call5.txt
: > $O
for v in sprio2 sprio3; do
  echo "##### $v" >> $O
  for m in 0 3; do for n in 32 64; do WN_V3_MODE=$m WN_DEV_LIB=tools/variants/libwn_$v.so timeout 120 python tools/rate.py cfg3 $n 2000 2 2>&1 | grep "samples/s" | sed "s/^/mode $m: /" >> $O; done; done
done
echo "=== anatomy x64 mode 3 (product build: skip group polls)" >> $O; WN_V3_MODE=3 timeout 150 python tools/profile_chain.py cfg3 64 2>&1 | grep -v amdgpu | grep "loop period\|multi\|skip group\|hand-off x\|ring tail\|layers>0" | cut -c1-500 >> $O
cat $O
