#!/bin/bash
mkdir -p gpurun_out
cd /root/repo
O=gpurun_out/r02_v3_forms_skip_packed.txt
: > $O
chk() { echo "## $5 WN_V3_MODE=$1 quick_check $2 x$3 N=$4" >> $O; WN_DEV_LIB=$5 WN_V3_MODE=$1 timeout 150 python tools/quick_check.py $2 $3 $4 2>&1 | grep -v amdgpu | tail -3 >> $O; }
rate() { echo "## $3 WN_V3_MODE=$1 rate x$2" >> $O; WN_DEV_LIB=$3 WN_V3_MODE=$1 timeout 150 python tools/rate.py cfg3 $2 2000 2 2>&1 | grep -v amdgpu | tail -2 >> $O; }
chk 0 cfg3 7 300 ""
chk 3 cfg3 64 600 ""
chk 3 cfg3 8 1300 tools/variants/libwn_qde2.so
rate 0 64 ""
for n in 64 96 128; do rate 3 $n ""; rate 3 $n tools/variants/libwn_qde2.so; done
rate 0 16 ""
cat $O
