#!/bin/bash
# round 3, GPU call 3: finer timing ablations in the two-streams-per-item form (results wrong by construction)
mkdir -p gpurun_out
cd /root/repo
O=gpurun_out/r03_call3.txt
: > $O
for a in 4 8 16 32 48; do
  echo "##### WN_V3_ABL=$a (4 skip group without dot, 8 skip group without loads/stores, 16 queue group without tap-0 dot, 32 without push, 48 both)" >> $O
  for n in 64; do WN_DEV_LIB=tools/variants/libwn_abl$a.so timeout 120 python tools/rate.py cfg3 $n 2000 3 2>&1 | grep "samples/s" >> $O; done
done
timeout 120 python tools/rate.py cfg3 64 2000 3 2>&1 | grep "samples/s" >> $O
cat $O
