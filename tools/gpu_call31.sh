#!/bin/bash
mkdir -p gpurun_out
cd /root/repo
O=gpurun_out/r02_v3_threshold.txt
: > $O
rate() { echo "## $3 WN_V3_MODE=$1 rate x$2" >> $O; WN_DEV_LIB=$3 WN_V3_MODE=$1 timeout 150 python tools/rate.py cfg3 $2 2000 2 2>&1 | grep -v amdgpu | tail -2 >> $O; }
rate 0 56 ""; rate 3 56 ""; rate 1 64 ""; rate 3 64 tools/variants/libwn_nopair.so; rate 0 1 tools/variants/libwn_nopair.so; rate 3 128 tools/variants/libwn_nopair.so
cat $O
