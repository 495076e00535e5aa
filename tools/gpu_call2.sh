#!/bin/bash
# GPU call 2 (round 2): first run of the wave-specialised kernel (variant 3)
mkdir -p gpurun_out
cd /root/repo
( time timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "wave_specialised or headline" 2>&1 | tail -40 ) > gpurun_out/r02_pytest_v3_first.log 2>&1
tail -5 gpurun_out/r02_pytest_v3_first.log
for ns in 16 32 64; do
  echo "=== v3 cfg3 x$ns" >> gpurun_out/r02_anatomy_v3.txt
  timeout 300 python tools/profile_chain.py cfg3 $ns >> gpurun_out/r02_anatomy_v3.txt 2>&1
done
grep -v amdgpu.ids gpurun_out/r02_anatomy_v3.txt | grep -v "^  layer [1-4]"
( time timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -40 ) > gpurun_out/r02_pytest_gpu_2.log 2>&1
tail -8 gpurun_out/r02_pytest_gpu_2.log
( time timeout 900 python bench.py --no-cpu-baseline ) > gpurun_out/r02_bench_2.json 2> gpurun_out/r02_bench_2.err
head -c 1500 gpurun_out/r02_bench_2.json
