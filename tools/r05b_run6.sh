#!/bin/bash
# round 5, second session, GPU run 6: cfg3 x64 against the number of sampler workgroups (env switch, product library)
mkdir -p gpurun_out
O=gpurun_out/r05b_run6.txt
: > $O
for s in 4 8 2 16 4; do
  echo "== WN_SAMPLERS=$s" >> $O
  WN_TESTING=1 WN_SAMPLERS=$s timeout 300 python tools/rate.py cfg3 64 3000 2 2>&1 | grep "samples/s" >> $O
done
cat $O
