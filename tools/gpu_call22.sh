#!/bin/bash
# two streams per pipeline item (WN_V3_MODE): oracle checks, then rates
mkdir -p gpurun_out
cd /root/repo
O=gpurun_out/r02_v3_g2_first.txt
: > $O
chk() { echo "## WN_V3_MODE=$1 quick_check $2 x$3 N=$4" >> $O; WN_V3_MODE=$1 timeout 150 python tools/quick_check.py $2 $3 $4 2>&1 | grep -v amdgpu | tail -3 >> $O; }
rate() { echo "## WN_V3_MODE=$1 rate x$2" >> $O; WN_V3_MODE=$1 timeout 150 python tools/rate.py cfg3 $2 2000 2 2>&1 | grep -v amdgpu | tail -2 >> $O; }
chk 1 cfg3 2 120
chk 2 cfg3 2 120
chk 2 cfg3 8 1300
chk 1 cfg3 6 700
chk 2 cfg3 64 1100
for m in 0 1 2; do rate $m 64; done
for n in 32 48 96 128; do rate 0 $n; rate 2 $n; done
cat $O
