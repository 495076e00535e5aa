#!/bin/bash
# kernel trace of the fp32 step and of the bf16x3 step, grouped by (kernel, grid)
mkdir -p gpurun_out
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_x3
WN_TESTING=1 timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_x3 -o x3 -- python $ROOT/tools/bench_train.py 32 16000 --no-torch --only-fp32 --x3 --reps=3 > /tmp/x3.log 2>&1
cd $ROOT
{ grep -h 'ms / step' /tmp/x3.log; python tools/rocprof_dispatches.py $(find /tmp/prof_x3 -name "*.db" | head -1) 100000 60 --group | grep -v "^# columns"; } > gpurun_out/r05b_x3_profile.txt 2>&1
cat gpurun_out/r05b_x3_profile.txt | cut -c1-140
