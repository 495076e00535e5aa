#!/bin/bash
mkdir -p gpurun_out
cd /root/repo
O=gpurun_out/r02_v3_rounds.txt
: > $O
chk() { echo "## quick_check $1 x$2 N=$3" >> $O; timeout 200 python tools/quick_check.py $1 $2 $3 2>&1 | grep -v amdgpu | tail -3 >> $O; }
rate() { echo "## rate x$1" >> $O; timeout 200 python tools/rate.py cfg3 $1 2000 1 2>&1 | grep -v amdgpu | tail -2 >> $O; }
chk cfg3 192 120
chk cfg3 301 100
for n in 160 192 256 512; do rate $n; done
cat $O
