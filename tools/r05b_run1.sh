#!/bin/bash
# round 5, second session, GPU run 1: the skip path's gradients with bf16-stored dzg and the bf16 shadow of dskip --
# parity (training + forward suites), then A/B of the config-5 bf16 step against the forms they replace, then a kernel trace.
mkdir -p gpurun_out
O=gpurun_out/r05b_run1.txt
: > $O
timeout 1200 python -m pytest tests/test_gpu_training.py tests/test_gpu_forward.py -m gpu -q 2>&1 | tail -12 >> $O
echo "== A/B (config-5 step, bf16 operands, 10 timed steps each, interleaved twice)" >> $O
run() {  # label, lib, env...
  local label=$1 lib=$2; shift 2
  echo "-- $label" >> $O
  env WN_TESTING=1 ${lib:+WN_DEV_LIB=$lib} "$@" timeout 300 python tools/bench_train.py 32 16000 --no-torch --only-bf16 --reps=10 2>&1 | grep "ms / step" >> $O
}
for rep in 1 2; do
  run "new (dzg bf16 + dskip shadow)" ""
  run "old (dzg fp32, no shadow)" tools/variants/libwn_dzg32.so WN_NO_DSKIP_SHADOW=1
  run "dzg bf16 only (no shadow)" "" WN_NO_DSKIP_SHADOW=1
  run "shadow only (dzg fp32)" tools/variants/libwn_dzg32.so
done
run "new, one stream" "" WN_TRAIN_ONE_STREAM=1
echo "== fp32 step" >> $O
env WN_TESTING=1 timeout 300 python tools/bench_train.py 32 16000 --no-torch --only-fp32 --reps=4 2>&1 | grep "ms / step" >> $O
cat $O
timeout 500 bash tools/profile_train.sh r05b_new > /dev/null 2>&1
head -c 3000 gpurun_out/rocprofv3_train_r05b_new.txt
