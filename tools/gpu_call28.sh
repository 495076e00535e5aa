#!/bin/bash
mkdir -p gpurun_out
cd /root/repo
( time timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 ) 2>&1 | tee gpurun_out/r02_pytest_gpu_forms2.txt
timeout 600 python bench.py 2>&1 | grep -v amdgpu | tail -1 > gpurun_out/r02_bench_forms2.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r02_bench_forms2.json"))
print("value", d["value"], "engine", d["engine_level"]["value"], "roofline", d["roofline"]["frac"], "verified", d["verified"], "x1", d["extra"]["cfg3x1"]["samples_per_s"], "x128", d["extra"]["cfg3x128"], "train bf16", d["extra"]["train_cfg5_bf16"]["ms_per_step"])
PY
bash tools/collect_profiles.sh forms2 > /dev/null 2>&1
head -c 2500 gpurun_out/rocprofv3_forms2.txt
