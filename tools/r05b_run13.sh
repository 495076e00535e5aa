#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/r05b_run13.txt
: > $O
for w in 1024 1536 2048 3072; do
  echo "-- fp32 WN_TN_WANT=$w" >> $O
  env WN_TESTING=1 WN_TN_WANT=$w timeout 300 python tools/bench_train.py 32 16000 --no-torch --only-fp32 --reps=4 2>&1 | grep "ms / step" >> $O
done
cat $O
