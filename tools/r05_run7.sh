#!/bin/bash
# round 5, seventh GPU call: training tests after the vectorised last-layer gate derivative, the training step's rocprofv3 passes on the step the bench times
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_training.py tests/test_gpu_forward.py -q -x > gpurun_out/r05_pytest_training.txt 2>&1
tail -n 4 gpurun_out/r05_pytest_training.txt
timeout 600 python bench.py --workload train5 --steps 8 --warmup 2 > gpurun_out/r05_bench_train5_b.json 2> gpurun_out/r05_bench_train5_b.err
cut -c1-300 gpurun_out/r05_bench_train5_b.json
timeout 1500 bash tools/collect_train_profiles.sh r05 bf16 > gpurun_out/r05_train_prof.log 2>&1
head -n 40 gpurun_out/rocprofv3_train_bf16_r05.txt | cut -c1-140
tail -n 14 gpurun_out/rocprofv3_train_bf16_r05.txt
