#!/bin/bash
mkdir -p gpurun_out
cd /root/repo
out=gpurun_out/r02_v3_io_group.txt
: > $out
timeout 120 python tools/quick_check.py cfg3 7 >> $out 2>&1
for ns in 16 32 48 64 96; do timeout 200 python tools/rate.py cfg3 $ns 2000 2 2>&1 | grep -v amdgpu.ids >> $out; done
for ns in 32 64; do
echo "=== v3 anatomy x$ns" >> $out
timeout 200 python tools/profile_chain.py cfg3 $ns 2>&1 | grep -v "amdgpu.ids\|^  layer [1-4]\|sampler [1-9]" >> $out
done
cat $out
( time timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "wave_specialised or headline" 2>&1 | tail -5 ) 2>&1
