#!/bin/bash
mkdir -p gpurun_out
cd /root/repo
timeout 300 python bench.py --samples 16000 --steps 2 --no-extra --no-cpu-baseline 2>&1 | grep -v amdgpu | tail -1 > gpurun_out/r02_bench_cfg3x64_16000_samples.json
timeout 300 python bench.py --scaling strong --samples 1000 --steps 2 --no-extra --no-cpu-baseline 2>&1 | grep -v amdgpu | tail -1 > gpurun_out/r02_bench_strong_512_streams_1gpu.json
python - <<'PY'
import json
for f in ("gpurun_out/r02_bench_cfg3x64_16000_samples.json", "gpurun_out/r02_bench_strong_512_streams_1gpu.json"):
    d = json.load(open(f))
    print(f, d["value"], d["ms_per_step"], d["verified"], d.get("engine_level", {}).get("value"), d["config"].get("chain"))
PY
