#!/bin/bash
mkdir -p gpurun_out
cd /root/repo
O=gpurun_out/r03_skip_nt.txt
: > $O
echo "# cache policy of the skip-lane hand-offs (WN_V3_SKIP_NT: 1 non-temporal stores, 2 non-temporal stores and loads)" >> $O
for n in 64 96 128; do timeout 50 python tools/rate.py cfg3 $n 3000 2 2>&1 | grep "samples/s" | sed "s/^/product: /" >> $O; done
for v in nt1 nt2; do
  WN_DEV_LIB=tools/variants/libwn_$v.so timeout 50 python tools/quick_check.py cfg3 64 2>&1 | grep quick_check | sed "s/^/$v: /" >> $O
  for n in 1 32 64 96 128; do WN_DEV_LIB=tools/variants/libwn_$v.so timeout 50 python tools/rate.py cfg3 $n 3000 2 2>&1 | grep "samples/s" | sed "s/^/$v: /" >> $O; done
done
cat $O
