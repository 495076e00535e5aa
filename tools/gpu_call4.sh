#!/bin/bash
mkdir -p gpurun_out
cd /root/repo
out=gpurun_out/r02_v3_three_groups.txt
: > $out
timeout 120 python tools/quick_check.py cfg3 7 >> $out 2>&1
timeout 120 python tools/quick_check.py cfg2 5 >> $out 2>&1
for ns in 16 32 64; do
  echo "=== v3 (C/S/Q groups) cfg3 x$ns" >> $out
  timeout 200 python tools/profile_chain.py cfg3 $ns 2>&1 | grep -v "amdgpu.ids\|^  layer [1-4]" >> $out
done
cat $out
( time timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "wave_specialised or headline or register_resident or baseline_configs" 2>&1 | tail -15 ) 2>&1 | tee gpurun_out/r02_pytest_v3_second.log
