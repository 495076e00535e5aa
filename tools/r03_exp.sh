#!/bin/bash
mkdir -p gpurun_out
cd /root/repo
O=gpurun_out/r03_presleep.txt
: > $O
echo "# pre-poll sleep of the critical group's polling waves (WN_V3_PRESLEEP eighths of the predicted wait; all variants with the last layer's skip group at priority 3)" >> $O
echo "## product build (no sleep)" >> $O; for n in 1 16 32 48 64 96 128; do timeout 120 python tools/rate.py cfg3 $n 3000 2 2>&1 | grep "samples/s" >> $O; done
for v in 3 4 5 6; do
  echo "## WN_V3_PRESLEEP=$v" >> $O
  for n in 1 64; do WN_DEV_LIB=tools/variants/libwn_ps$v.so timeout 120 python tools/quick_check.py cfg3 $n 2>&1 | grep quick_check >> $O; done
  for n in 1 16 32 48 64 96 128; do WN_DEV_LIB=tools/variants/libwn_ps$v.so timeout 120 python tools/rate.py cfg3 $n 3000 2 2>&1 | grep "samples/s" >> $O; done
done
cat $O
