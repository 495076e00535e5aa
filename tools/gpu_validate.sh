#!/bin/bash
# Runs on the GPU box: the GPU test suite, smoke(), the default bench line.   tools/gpu_validate.sh [tag]
TAG=${1:-latest}
mkdir -p gpurun_out
cd /root/repo
( time timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 ) 2>&1 | tee gpurun_out/pytest_gpu_$TAG.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu | tail -2
( time timeout 900 python bench.py 2> gpurun_out/bench_$TAG.err | tail -1 > gpurun_out/bench_$TAG.json ) 2>&1 | tail -3
tail -5 gpurun_out/bench_$TAG.err
python - <<PY
import json
d = json.load(open("gpurun_out/bench_$TAG.json"))
e = d.get("extra", {})
print("value", d["value"], "median ms", d.get("median_ms_per_step"), "engine", d["engine_level"]["value"], "roofline", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"], d["roofline"].get("traffic_kind", "")[:40], d["roofline"].get("traffic_over_algorithmic"), "verified", d["verified"])
for k, v in e.items():
    print(" ", k, {kk: v[kk] for kk in v if kk in ("samples_per_s", "hbm_frac", "kernel_variant", "verified", "ms_per_step", "mfma_peak_frac", "batched_ms", "error")})
print("cpu", d.get("cpu_baseline", {}).get("matrix"))
PY
