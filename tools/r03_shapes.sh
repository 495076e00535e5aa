#!/bin/bash
mkdir -p gpurun_out
cd /root/repo
O=gpurun_out/r03_shapes.txt
: > $O
for c in "cfg2 1" "cfg2 4" "cfg2 64" "cfg1 1" "cfg1 3" "cfg1 64" "cfg3 1" "cfg3 64"; do set -- $c; timeout 60 python tools/quick_check.py $1 $2 2>&1 | grep "quick_check\|Error\|error" | head -3 >> $O; done
WN_V3_MODE=3 timeout 60 python tools/quick_check.py cfg2 6 2>&1 | grep "quick_check\|Error" | sed "s/^/mode 3: /" >> $O
WN_V3_MODE=3 timeout 60 python tools/quick_check.py cfg1 6 2>&1 | grep "quick_check\|Error" | sed "s/^/mode 3: /" >> $O
for c in "cfg2 1" "cfg2 64" "cfg1 1" "cfg1 64"; do set -- $c; timeout 60 python tools/rate.py $1 $2 4000 2 2>&1 | grep "samples/s" >> $O; done
cat $O
