#!/bin/bash
# round 5, second session, GPU run 7: the training step's products as three bf16 planes per operand (mode 2, "bf16x3") -- parity on the fp32 step's fixtures and bars, timing
mkdir -p gpurun_out
O=gpurun_out/r05b_run7.txt
: > $O
timeout 900 python -m pytest tests/test_gpu_training.py -m gpu -q -k "headline_widths or reference_golden" 2>&1 | tail -15 >> $O
echo "== config-5 step" >> $O
env WN_TESTING=1 timeout 300 python tools/bench_train.py 32 16000 --no-torch --only-fp32 --x3 --reps=4 2>&1 | grep "ms / step" >> $O
cat $O
