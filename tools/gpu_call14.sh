#!/bin/bash
mkdir -p gpurun_out
cd /root/repo
out=gpurun_out/r02_v3_poll_variants.txt
: > $out
for v in base ss2 ss2cs1 ss6; do
  echo "##### $v" >> $out
  WN_DEV_LIB=tools/variants/libwn_$v.so timeout 120 python tools/quick_check.py cfg3 7 2>&1 | grep quick >> $out
  for ns in 32 48 64; do WN_DEV_LIB=tools/variants/libwn_$v.so timeout 200 python tools/rate.py cfg3 $ns 2000 2 2>&1 | grep -v amdgpu.ids >> $out; done
done
cat $out
