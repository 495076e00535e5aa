#!/bin/bash
# round 5, second session, GPU run 10: the row splits of the weight-gradient products against the resident workgroup slots
mkdir -p gpurun_out
O=gpurun_out/r05b_run10.txt
: > $O
run() {  # label, env...
  local label=$1; shift
  echo "-- $label" >> $O
  env WN_TESTING=1 "$@" timeout 300 python tools/bench_train.py 32 16000 --no-torch --only-bf16 --reps=10 2>&1 | grep "ms / step" >> $O
}
for rep in 1 2; do
  run "product (1024 / 512)"
  run "WN_TN_WANT=768" WN_TN_WANT=768
  run "WN_TN_WANT=512" WN_TN_WANT=512
  run "WN_TN_WANT=1536" WN_TN_WANT=1536
  run "WN_TN_WANT_WIDE=1024" WN_TN_WANT_WIDE=1024
  run "WN_TN_WANT_WIDE=256" WN_TN_WANT_WIDE=256
done
cat $O
