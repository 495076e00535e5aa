#!/bin/bash
mkdir -p gpurun_out
cd /root/repo
O=gpurun_out/r03_facade_overlap.txt
: > $O
( timeout 600 python -m pytest tests/test_gpu_facade.py -m gpu -q -x 2>&1 | tail -2 ) >> $O 2>&1
timeout 300 python bench.py --no-extra --no-cpu-baseline --steps 4 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('facade', d['value'], 'ms', d['ms_per_step'], 'median', d['median_ms_per_step'], 'engine', d['engine_level']['value'], 'ratio', d['engine_level']['facade_over_engine'], 'verified', d['verified'])" >> $O
cat $O
