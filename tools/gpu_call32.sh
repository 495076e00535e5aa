#!/bin/bash
mkdir -p gpurun_out
cd /root/repo
O=gpurun_out/r02_v3_split_residual.txt
: > $O
chk() { echo "## WN_V3_MODE=$1 quick_check $2 x$3 N=$4" >> $O; WN_V3_MODE=$1 timeout 150 python tools/quick_check.py $2 $3 $4 2>&1 | grep -v amdgpu | tail -3 >> $O; }
rate() { echo "## WN_V3_MODE=$1 rate x$2" >> $O; WN_V3_MODE=$1 timeout 150 python tools/rate.py cfg3 $2 2000 2 2>&1 | grep -v amdgpu | tail -2 >> $O; }
chk 3 cfg3 64 600
chk 1 cfg3 6 700
rate 3 64; rate 3 56; rate 3 96; rate 3 128
cat $O
