#!/bin/bash
# round 5, second session, GPU run 5: timing ablation -- the fused layer kernels without their weight re-reads (results wrong)
mkdir -p gpurun_out
O=gpurun_out/r05b_run5.txt
: > $O
run() {  # label, lib, env...
  local label=$1 lib=$2; shift 2
  echo "-- $label" >> $O
  env WN_TESTING=1 ${lib:+WN_DEV_LIB=$lib} "$@" timeout 300 python tools/bench_train.py 32 16000 --no-torch --only-bf16 --reps=10 2>&1 | grep "ms / step" >> $O
}
for rep in 1 2; do
  run "product" ""
  run "ablation: weight chunks read once per tile" tools/variants/libwn_now.so
done
cat $O
WN_TESTING=1 WN_DEV_LIB=tools/variants/libwn_now.so timeout 500 bash tools/profile_train.sh r05b_now > /dev/null 2>&1
head -c 1500 gpurun_out/rocprofv3_train_r05b_now.txt
