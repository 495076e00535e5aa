#!/bin/bash
mkdir -p gpurun_out
cd /root/repo
O=gpurun_out/r03_shapes4.txt
: > $O
for c in "chaconne 1" "chaconne 5" "chaconne 40" "cfg2 1" "cfg3 1"; do set -- $c; timeout 60 python tools/quick_check.py $1 $2 2>&1 | grep "quick_check\|Error\|error" | head -3 >> $O; done
for c in "chaconne 1" "chaconne 32" "chaconne 64"; do set -- $c; timeout 60 python tools/rate.py $1 $2 4000 2 2>&1 | grep "samples/s" >> $O; done
WN_KERNEL=v2 timeout 60 python tools/rate.py chaconne 1 4000 2 2>&1 | grep "samples/s" | sed "s/^/v2: /" >> $O
WN_KERNEL=v2 timeout 60 python tools/rate.py chaconne 64 4000 2 2>&1 | grep "samples/s" | sed "s/^/v2: /" >> $O
cat $O
