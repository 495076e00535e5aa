#!/bin/bash
# round 5, run 2: full GPU suite on the new build (tap request behind barrier A in the two-streams form, residency barrier, new gate), rates, bench
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r05_pytest_gpu_1.txt
out=gpurun_out/r05_run2.txt; : > $out
V=tools/variants
for lib in product cur_fast; do
  echo "== $lib" >> $out
  if [ $lib = product ]; then unset WN_DEV_LIB; else export WN_DEV_LIB=$V/libwn_$lib.so; fi
  for s in 64 1 32 128; do timeout 300 python tools/rate.py cfg3 $s 3000 2 2>&1 | grep "samples/s" >> $out; done
done
for lib in product cur_fast; do
  echo "== anatomy $lib x1" >> $out
  if [ $lib = product ]; then unset WN_DEV_LIB; else export WN_DEV_LIB=$V/libwn_$lib.so; fi
  timeout 300 python tools/profile_chain.py cfg3 1 2>&1 | tail -75 >> $out
done
unset WN_DEV_LIB
timeout 900 python bench.py > gpurun_out/r05_bench_1.json 2> gpurun_out/r05_bench_1.err
cat gpurun_out/r05_pytest_gpu_1.txt; cat $out | head -30; cat gpurun_out/r05_bench_1.json | head -c 3000
