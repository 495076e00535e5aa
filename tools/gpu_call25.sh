#!/bin/bash
mkdir -p gpurun_out
cd /root/repo
O=gpurun_out/r02_v3_forms_samplers.txt
: > $O
rate() { echo "## WN_V3_MODE=$1 WN_SAMPLERS=$3 rate x$2" >> $O; WN_SAMPLERS=$3 WN_V3_MODE=$1 timeout 150 python tools/rate.py cfg3 $2 2000 2 2>&1 | grep -v amdgpu | tail -2 >> $O; }
for k in 6 8; do for n in 64 96 128; do rate 3 $n $k; done; done
rate 3 144 8
rate 3 112 8
rate 1 128 8
echo "##### WN_V3_MODE=3 WN_SAMPLERS=8 anatomy x128" >> $O; WN_SAMPLERS=8 WN_V3_MODE=3 timeout 150 python tools/profile_chain.py cfg3 128 2>&1 | grep -v amdgpu | cut -c1-260 >> $O
cat $O
