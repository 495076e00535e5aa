#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/r05b_run11.txt
: > $O
run() {  # label, env...
  local label=$1; shift
  echo "-- $label" >> $O
  env WN_TESTING=1 "$@" timeout 300 python tools/bench_train.py 32 16000 --no-torch --only-bf16 --reps=10 2>&1 | grep "ms / step" >> $O
}
for rep in 1 2; do
  run "WN_TN_WANT=512" WN_TN_WANT=512
  run "WN_TN_WANT=384" WN_TN_WANT=384
  run "WN_TN_WANT=256" WN_TN_WANT=256
  run "WN_TN_WANT=640" WN_TN_WANT=640
  run "WN_TN_WANT=512 WIDE=384" WN_TN_WANT=512 WN_TN_WANT_WIDE=384
  run "WN_TN_WANT=512 WIDE=768" WN_TN_WANT=512 WN_TN_WANT_WIDE=768
done
echo "== fp32 step" >> $O
for w in 1024 512 256; do
  echo "-- fp32 WN_TN_WANT=$w" >> $O
  env WN_TESTING=1 WN_TN_WANT=$w timeout 300 python tools/bench_train.py 32 16000 --no-torch --only-fp32 --reps=4 2>&1 | grep "ms / step" >> $O
done
cat $O
