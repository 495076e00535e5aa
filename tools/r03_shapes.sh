#!/bin/bash
mkdir -p gpurun_out
cd /root/repo
O=gpurun_out/r03_shapes3.txt
: > $O
( timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_facade.py -m gpu -q -x 2>&1 | tail -5 ) >> $O 2>&1
cat $O
