#!/bin/bash
mkdir -p gpurun_out
cd /root/repo
O=gpurun_out/r02_v3_queue_priority.txt
: > $O
rate() { echo "## $2 rate x$1" >> $O; WN_DEV_LIB=$2 timeout 150 python tools/rate.py cfg3 $1 2000 2 2>&1 | grep -v amdgpu | tail -2 >> $O; }
for n in 64 128; do rate $n ""; rate $n tools/variants/libwn_qprio1.so; rate $n tools/variants/libwn_qprio2.so; done
cat $O
