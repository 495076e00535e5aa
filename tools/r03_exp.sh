#!/bin/bash
mkdir -p gpurun_out
cd /root/repo
O=gpurun_out/r03_lds_queues.txt
: > $O
for c in cfg3 cfg2 cfg1 chaconne; do timeout 50 python tools/quick_check.py $c 1 2>&1 | grep quick_check >> $O; done
for c in cfg3 cfg3 cfg2 cfg1 chaconne; do timeout 50 python tools/rate.py $c 1 6000 2 2>&1 | grep "samples/s" >> $O; done
( timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_facade.py -m gpu -q -x -k "ns1 or full_size or export_queue or queue_state or golden or host_calls or baseline_configs or batched_priming" 2>&1 | tail -3 ) >> $O 2>&1
cat $O
