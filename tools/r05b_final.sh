#!/bin/bash
# round 5, second session: the tree as committed -- full GPU suite, smoke, bench line (default), train5 line, PMC passes of the training step
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -6 > gpurun_out/r05b_pytest_gpu_final.txt
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -2 >> gpurun_out/r05b_pytest_gpu_final.txt
timeout 1200 python bench.py > gpurun_out/r05b_bench_final_gen.json 2> gpurun_out/r05b_bench_final_gen.err
timeout 600 python bench.py --workload train5 --steps 5 --warmup 2 > gpurun_out/r05b_bench_final_train5.json 2> gpurun_out/r05b_bench_final_train5.err

cat gpurun_out/r05b_pytest_gpu_final.txt | cut -c1-300; head -c 400 gpurun_out/r05b_bench_final_gen.json; echo; head -c 500 gpurun_out/r05b_bench_final_train5.json; echo
