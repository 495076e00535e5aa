#!/bin/bash
# GPU call 1 (round 2): full GPU suite, bench line, token-count sweep of the one-chain kernel
mkdir -p gpurun_out
cd /root/repo
( time timeout 1500 python -m pytest tests -m gpu -x -q -s 2>&1 | tail -60 ) > gpurun_out/r02_pytest_gpu_1.log 2>&1
( time timeout 900 python bench.py ) > gpurun_out/r02_bench_1.json 2> gpurun_out/r02_bench_1.err
for ns in 16 24 32 40 48 64; do
  echo "=== WN_CHAINS=1 cfg3 x$ns" >> gpurun_out/r02_sweep_one_chain.txt
  WN_CHAINS=1 timeout 300 python tools/profile_chain.py cfg3 $ns >> gpurun_out/r02_sweep_one_chain.txt 2>&1
done
echo "=== default (two chains) cfg3 x64" >> gpurun_out/r02_sweep_one_chain.txt
timeout 300 python tools/profile_chain.py cfg3 64 >> gpurun_out/r02_sweep_one_chain.txt 2>&1
tail -5 gpurun_out/r02_pytest_gpu_1.log
cat gpurun_out/r02_bench_1.json | head -c 3000
