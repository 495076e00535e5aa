#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/r05b_run14.txt
: > $O
for rep in 1 2; do
  echo "-- product" >> $O
  env WN_TESTING=1 timeout 300 python tools/bench_train.py 32 16000 --no-torch --only-bf16 --reps=10 2>&1 | grep "ms / step" >> $O
  echo "-- ablation: plain stores instead of the weight gradients' atomics" >> $O
  env WN_TESTING=1 WN_DEV_LIB=tools/variants/libwn_tnplain.so timeout 300 python tools/bench_train.py 32 16000 --no-torch --only-bf16 --reps=10 2>&1 | grep "ms / step" >> $O
done
cat $O
