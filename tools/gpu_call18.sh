#!/bin/bash
mkdir -p gpurun_out
cd /root/repo
( time timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -25 ) > gpurun_out/r02_pytest_gpu_3.log 2>&1
tail -6 gpurun_out/r02_pytest_gpu_3.log
( time timeout 900 python bench.py ) > gpurun_out/r02_bench_3.json 2> gpurun_out/r02_bench_3.err
head -c 2500 gpurun_out/r02_bench_3.json; echo
bash tools/collect_profiles.sh r02_v3 > /dev/null 2>&1
cat gpurun_out/rocprofv3_r02_v3.txt | cut -c1-220
for ns in 16 64; do
echo "=== v3 anatomy x$ns" >> gpurun_out/r02_chain_anatomy_v3.txt
timeout 200 python tools/profile_chain.py cfg3 $ns 2>&1 | grep -v "amdgpu.ids" >> gpurun_out/r02_chain_anatomy_v3.txt
done
