#!/bin/bash
mkdir -p gpurun_out
cd /root/repo
out=gpurun_out/r02_v3_small_ns.txt
: > $out
for ns in 1 2 3; do timeout 120 python tools/quick_check.py cfg3 $ns 200 >> $out 2>&1; done
for ns in 1 2 4 8; do timeout 200 python tools/rate.py cfg3 $ns 8000 2 2>&1 | grep -v amdgpu.ids >> $out; done
WN_KERNEL=v2 timeout 200 python tools/rate.py cfg3 1 8000 2 2>&1 | grep -v amdgpu.ids >> $out
grep -v amdgpu $out
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_facade.py -m gpu -q -x 2>&1 | tail -6 ) 2>&1
