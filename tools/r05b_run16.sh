#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/r05b_run16.txt
: > $O
timeout 1200 python -m pytest tests/test_gpu_training.py tests/test_gpu_forward.py -m gpu -q 2>&1 | tail -3 >> $O
for rep in 1 2 3; do
  echo "-- skip products enqueued block by block" >> $O
  env WN_TESTING=1 timeout 300 python tools/bench_train.py 32 16000 --no-torch --only-bf16 --reps=10 2>&1 | grep "ms / step" >> $O
  echo "-- all up front (rounds 2-4)" >> $O
  env WN_TESTING=1 WN_TRAIN_SKIP_UPFRONT=1 timeout 300 python tools/bench_train.py 32 16000 --no-torch --only-bf16 --reps=10 2>&1 | grep "ms / step" >> $O
done
echo "-- fp32" >> $O
env WN_TESTING=1 timeout 300 python tools/bench_train.py 32 16000 --no-torch --only-fp32 --reps=4 2>&1 | grep "ms / step" >> $O
echo "-- fp32, all up front" >> $O
env WN_TESTING=1 WN_TRAIN_SKIP_UPFRONT=1 timeout 300 python tools/bench_train.py 32 16000 --no-torch --only-fp32 --reps=4 2>&1 | grep "ms / step" >> $O
cat $O
