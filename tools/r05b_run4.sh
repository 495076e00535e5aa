#!/bin/bash
# round 5, second session, GPU run 4: the gate pair recomputed in the fused backward kernel (wn_bwd_layer_bf16<true>) instead of stored --
# parity (training + forward suites), then A/B against the stored pair (WN_NO_GATE_RECOMPUTE=1, same library).
mkdir -p gpurun_out
O=gpurun_out/r05b_run4.txt
: > $O
timeout 1200 python -m pytest tests/test_gpu_training.py tests/test_gpu_forward.py -m gpu -q 2>&1 | tail -25 >> $O
run() {  # label, lib, env...
  local label=$1 lib=$2; shift 2
  echo "-- $label" >> $O
  env WN_TESTING=1 ${lib:+WN_DEV_LIB=$lib} "$@" timeout 300 python tools/bench_train.py 32 16000 --no-torch --only-bf16 --reps=10 2>&1 | grep "ms / step" >> $O
}
for rep in 1 2 3; do
  run "recomputed gate pair" ""
  run "stored gate pair" "" WN_NO_GATE_RECOMPUTE=1
done
cat $O
timeout 500 bash tools/profile_train.sh r05b_rc > /dev/null 2>&1
head -c 2000 gpurun_out/rocprofv3_train_r05b_rc.txt
