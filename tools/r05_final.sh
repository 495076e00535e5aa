#!/bin/bash
# round 5: the tree as committed -- full GPU suite, smoke, bench line (default), train5 line
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -6 > gpurun_out/r05_pytest_gpu_final2.txt
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -2 >> gpurun_out/r05_pytest_gpu_final2.txt
timeout 1200 python bench.py > gpurun_out/r05_bench_3.json 2> gpurun_out/r05_bench_3.err
timeout 600 python bench.py --workload train5 --steps 5 --warmup 2 > gpurun_out/r05_bench_train5.json 2> gpurun_out/r05_bench_train5.err
cat gpurun_out/r05_pytest_gpu_final2.txt; head -c 600 gpurun_out/r05_bench_3.json; echo; head -c 900 gpurun_out/r05_bench_train5.json; tail -3 gpurun_out/r05_bench_3.err gpurun_out/r05_bench_train5.err
