#!/bin/bash
mkdir -p gpurun_out
cd /root/repo
O=gpurun_out/r02_v3_g2_anatomy.txt
: > $O
for m in 1 2; do echo "##### WN_V3_MODE=$m anatomy x64" >> $O; WN_V3_MODE=$m timeout 150 python tools/profile_chain.py cfg3 64 2>&1 | grep -v amdgpu >> $O; done
rate() { echo "## WN_V3_MODE=$1 rate x$2" >> $O; WN_V3_MODE=$1 timeout 150 python tools/rate.py cfg3 $2 2000 2 2>&1 | grep -v amdgpu | tail -2 >> $O; }
rate 1 96
rate 1 128
rate 1 48
cat $O
