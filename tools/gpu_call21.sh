#!/bin/bash
mkdir -p gpurun_out
cd /root/repo
( time timeout 1200 python -m pytest tests/test_gpu_training.py -m gpu -q -x 2>&1 | tail -8 ) 2>&1 | tee gpurun_out/r02_pytest_training.log
python - <<'PY' 2>&1 | grep -v amdgpu | tee gpurun_out/r02_train_cfg5.txt
import sys, json
sys.path.insert(0, "/root/repo")
import bench
for prec in ("fp32", "bf16"):
    print(prec, json.dumps(bench.train_step_cfg5(0, precision=prec)))
PY
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_tr
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_tr -o tr -- python -c "
import sys; sys.path.insert(0, '/root/repo')
import bench
print(bench.train_step_cfg5(0, precision='bf16', reps=3))
" > /tmp/tr.log 2>&1
cd /root/repo
python tools/rocprof_summary.py $(find /tmp/prof_tr -name "*.db") 2>&1 | head -16 | cut -c1-150 | tee gpurun_out/r02_rocprofv3_train_step_cfg5_bf16.txt
