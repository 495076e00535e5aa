#!/bin/bash
mkdir -p gpurun_out
cd /root/repo
O=gpurun_out/r03_skip_two_chains.txt
: > $O
for n in 1 7 64; do timeout 50 python tools/quick_check.py cfg3 $n 2>&1 | grep quick_check >> $O; done
timeout 50 python tools/quick_check.py cfg2 5 2>&1 | grep quick_check >> $O
for n in 1 32 64 64 96 128; do timeout 50 python tools/rate.py cfg3 $n 3000 2 2>&1 | grep "samples/s" >> $O; done
cat $O
