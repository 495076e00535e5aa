#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/r05b_run15.txt
: > $O
timeout 1200 python -m pytest tests/test_gpu_training.py tests/test_gpu_forward.py -m gpu -q 2>&1 | tail -3 >> $O
for rep in 1 2 3; do
  echo "-- epilogue roles from an opaque index (stand-alone bf16 products)" >> $O
  env WN_TESTING=1 timeout 300 python tools/bench_train.py 32 16000 --no-torch --only-bf16 --reps=10 2>&1 | grep "ms / step" >> $O
  echo "-- before" >> $O
  env WN_TESTING=1 WN_DEV_LIB=tools/variants/libwn_prev.so timeout 300 python tools/bench_train.py 32 16000 --no-torch --only-bf16 --reps=10 2>&1 | grep "ms / step" >> $O
done
echo "-- fp32" >> $O
env WN_TESTING=1 timeout 300 python tools/bench_train.py 32 16000 --no-torch --only-fp32 --reps=4 2>&1 | grep "ms / step" >> $O
env WN_TESTING=1 WN_DEV_LIB=tools/variants/libwn_prev.so timeout 300 python tools/bench_train.py 32 16000 --no-torch --only-fp32 --reps=4 2>&1 | grep "ms / step" >> $O
cat $O
