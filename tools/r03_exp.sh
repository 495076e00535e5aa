#!/bin/bash
mkdir -p gpurun_out
cd /root/repo
O=gpurun_out/r03_queue_pair_rows.txt
: > $O
for c in "cfg3 64" "cfg3 6" "cfg2 64" "cfg1 40" "cfg3 1"; do set -- $c; WN_V3_MODE=3 timeout 50 python tools/quick_check.py $1 $2 2>&1 | grep quick_check >> $O; done
for n in 64 64 96 128; do timeout 50 python tools/rate.py cfg3 $n 3000 2 2>&1 | grep "samples/s" >> $O; done
timeout 50 python tools/rate.py cfg2 64 3000 2 2>&1 | grep "samples/s" >> $O
cat $O
