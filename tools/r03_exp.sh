#!/bin/bash
mkdir -p gpurun_out
cd /root/repo
O=gpurun_out/r03_dedicated_fetcher.txt
: > $O
echo "# timing experiment (results wrong: no skip work at all, WN_V3_ABL=1): the critical group fetches its input (baseline) vs the idle skip group as a DEDICATED fetcher" >> $O
for v in abl1 fetch; do
  echo "## $v" >> $O
  for m in 0 3; do for n in 16 32 48 64 96 128; do WN_V3_MODE=$m WN_DEV_LIB=tools/variants/libwn_$v.so timeout 50 python tools/rate.py cfg3 $n 3000 1 2>&1 | grep "samples/s" | sed "s/^/mode $m: /" >> $O; done; done
done
cat $O
