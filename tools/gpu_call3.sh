#!/bin/bash
# GPU call 3 (round 2): variants of the wave-specialised kernel (priority x tail placement)
mkdir -p gpurun_out
cd /root/repo
out=gpurun_out/r02_v3_variants.txt
: > $out
for v in p1m0 p0m0 p1m1 p0m1; do
  echo "##### variant $v (PRIO=${v:1:1} TAILMODE=${v:3:1})" >> $out
  WN_DEV_LIB=tools/variants/libwn_$v.so timeout 120 python tools/quick_check.py cfg3 7 >> $out 2>&1
  for ns in 16 64; do
    echo "=== $v cfg3 x$ns" >> $out
    WN_DEV_LIB=tools/variants/libwn_$v.so timeout 200 python tools/profile_chain.py cfg3 $ns 2>&1 | grep -v "amdgpu.ids\|^  layer [1-4]" >> $out
  done
done
cat $out
