#!/bin/bash
# round 5, second session, GPU run 3: z of a skip block's layers side by side in one matrix (no second copy on the skip rows) --
# parity (training + forward suites), then A/B against the build before it.
mkdir -p gpurun_out
O=gpurun_out/r05b_run3.txt
: > $O
timeout 1200 python -m pytest tests/test_gpu_training.py tests/test_gpu_forward.py -m gpu -q 2>&1 | tail -4 >> $O
run() {  # label, lib, env...
  local label=$1 lib=$2; shift 2
  echo "-- $label" >> $O
  env WN_TESTING=1 ${lib:+WN_DEV_LIB=$lib} "$@" timeout 300 python tools/bench_train.py 32 16000 --no-torch --only-bf16 --reps=10 2>&1 | grep "ms / step" >> $O
}
for rep in 1 2 3; do
  run "new (Z_b: no zg copy)" ""
  run "before (zg copy)" tools/variants/libwn_s12.so
done
echo "== fp32 step" >> $O
env WN_TESTING=1 timeout 300 python tools/bench_train.py 32 16000 --no-torch --only-fp32 --reps=4 2>&1 | grep "ms / step" >> $O
env WN_TESTING=1 WN_DEV_LIB=tools/variants/libwn_s12.so timeout 300 python tools/bench_train.py 32 16000 --no-torch --only-fp32 --reps=4 2>&1 | grep "ms / step" >> $O
cat $O
timeout 500 bash tools/profile_train.sh r05b_s3 > /dev/null 2>&1
head -c 2200 gpurun_out/rocprofv3_train_r05b_s3.txt
