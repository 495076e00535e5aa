#!/bin/bash
mkdir -p gpurun_out
cd /root/repo
( time timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -4 ) 2>&1 | tee gpurun_out/r02_pytest_gpu_final2.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu | tail -1
timeout 600 python bench.py 2>&1 | grep -v amdgpu | tail -1 > gpurun_out/r02_bench_final2.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r02_bench_final2.json"))
print("value", d["value"], "engine", d["engine_level"], "roofline", d["roofline"]["frac"], "verified", d["verified"], "x1", d["extra"]["cfg3x1"]["samples_per_s"], "x128", d["extra"]["cfg3x128"]["samples_per_s"], "train bf16", d["extra"]["train_cfg5_bf16"]["ms_per_step"])
PY
