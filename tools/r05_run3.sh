#!/bin/bash
# round 5, run 3: the filter/gate window with fewer instructions (k-packed FMAs, exp2 gate), wait states in front of the polls' loads, rest of the GPU suite
mkdir -p gpurun_out
out=gpurun_out/r05_run3.txt; : > $out
V=tools/variants
for s in 64 7 1; do timeout 300 python tools/quick_check.py cfg3 $s 2>&1 | grep quick_check >> $out; done
timeout 300 python tools/quick_check.py cfg2 64 2>&1 | grep quick_check >> $out
timeout 300 python tools/quick_check.py chaconne 8 2>&1 | grep quick_check >> $out
for lib in product nop0 nop5 nop10 nop16 kpack0 gate0; do
  echo "== $lib" >> $out
  if [ $lib = product ]; then unset WN_DEV_LIB; else export WN_DEV_LIB=$V/libwn_$lib.so; fi
  for s in 64 1 32 128; do timeout 300 python tools/rate.py cfg3 $s 3000 2 2>&1 | grep "samples/s" >> $out; done
done
unset WN_DEV_LIB
echo "== anatomy product x64" >> $out
timeout 300 python tools/profile_chain.py cfg3 64 2>&1 | grep -v "^  layer\|^   L" >> $out
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/r05_pytest_gpu_2.txt
cat $out; cat gpurun_out/r05_pytest_gpu_2.txt
