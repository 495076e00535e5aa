#!/bin/bash
# forms of the wave-specialised kernel: WN_V3_MODE bit 0 = two streams per layer item, bit 1 = two head replicas
mkdir -p gpurun_out
cd /root/repo
O=gpurun_out/r02_v3_forms.txt
: > $O
chk() { echo "## WN_V3_MODE=$1 quick_check $2 x$3 N=$4" >> $O; WN_V3_MODE=$1 timeout 150 python tools/quick_check.py $2 $3 $4 2>&1 | grep -v amdgpu | tail -3 >> $O; }
rate() { echo "## WN_V3_MODE=$1 rate x$2" >> $O; WN_V3_MODE=$1 timeout 150 python tools/rate.py cfg3 $2 2000 2 2>&1 | grep -v amdgpu | tail -2 >> $O; }
chk 3 cfg3 6 700
chk 2 cfg3 7 300
chk 3 cfg3 64 1100
for m in 0 3; do rate $m 64; done
rate 2 64
for n in 80 96 128 192; do rate 3 $n; done
rate 0 1; rate 0 16
echo "##### WN_V3_MODE=3 anatomy x128" >> $O; WN_V3_MODE=3 timeout 150 python tools/profile_chain.py cfg3 128 2>&1 | grep -v amdgpu >> $O
cat $O
