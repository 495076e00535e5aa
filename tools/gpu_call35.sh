#!/bin/bash
mkdir -p gpurun_out
cd /root/repo
( time timeout 1200 python -m pytest tests/test_gpu_training.py tests/test_gpu_forward.py -m gpu -q -x 2>&1 | tail -8 ) 2>&1 | tee gpurun_out/r02_pytest_training_bf16_storage.log
python - <<'PY' 2>&1 | grep -v amdgpu | tee gpurun_out/r02_train_cfg5_bf16_storage.txt
import sys, json
sys.path.insert(0, "/root/repo")
import bench
for prec in ("fp32", "bf16"):
    print(prec, json.dumps(bench.train_step_cfg5(0, precision=prec)))
PY
