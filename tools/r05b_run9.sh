#!/bin/bash
# kernel trace of the bf16 step with every product on ONE stream: each kernel's stand-alone duration
mkdir -p gpurun_out
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_one
WN_TESTING=1 WN_TRAIN_ONE_STREAM=1 timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_one -o one -- python $ROOT/tools/bench_train.py 32 16000 --no-torch --only-bf16 --reps=3 > /tmp/one.log 2>&1
cd $ROOT
{ grep -h 'ms / step' /tmp/one.log; python tools/rocprof_dispatches.py $(find /tmp/prof_one -name "*.db" | head -1) 100000 40 --group | grep -v "^# columns"; } > gpurun_out/r05b_one_stream_profile.txt 2>&1
cat gpurun_out/r05b_one_stream_profile.txt | cut -c1-150
