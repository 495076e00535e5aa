#!/bin/bash
# round 5, sixth GPU call: the gate / residency / churn tests on the rebuilt library, then the training step's rocprofv3 passes with the fused optimiser
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "churn or thread or process or resident or gate or hog" > gpurun_out/r05_pytest_gate_churn.txt 2>&1
tail -n 5 gpurun_out/r05_pytest_gate_churn.txt
timeout 1500 bash tools/collect_train_profiles.sh r05 bf16 > gpurun_out/r05_train_prof.log 2>&1
tail -n 30 gpurun_out/rocprofv3_train_bf16_r05.txt
