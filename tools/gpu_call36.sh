#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_tr
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_tr -o tr -- python -c "
import sys; sys.path.insert(0, '/root/repo')
import bench
print(bench.train_step_cfg5(0, precision='bf16', reps=3))
" > /tmp/tr.log 2>&1
cd /root/repo
python tools/rocprof_summary.py $(find /tmp/prof_tr -name "*.db") 2>&1 | head -12 | cut -c1-150 | tee gpurun_out/r02_rocprofv3_train_bf16_storage.txt
