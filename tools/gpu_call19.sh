#!/bin/bash
mkdir -p gpurun_out
cd /root/repo
( time timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -12 ) > gpurun_out/r02_pytest_gpu_4.log 2>&1
tail -6 gpurun_out/r02_pytest_gpu_4.log
( time timeout 900 python bench.py ) > gpurun_out/r02_bench_4.json 2> gpurun_out/r02_bench_4.err
head -c 1500 gpurun_out/r02_bench_4.json; echo
