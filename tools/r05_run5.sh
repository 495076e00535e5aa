#!/bin/bash
# round 5, run 5: the build that ships -- full GPU suite, bench line, rocprofv3 stats + PMC passes, TCC passes, one stream per item at 64 streams with the tap request behind barrier A
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -12 > gpurun_out/r05_pytest_gpu_final.txt
timeout 1200 python bench.py > gpurun_out/r05_bench_2.json 2> gpurun_out/r05_bench_2.err
bash tools/collect_profiles.sh r05 > gpurun_out/r05_collect.log 2>&1
bash tools/collect_tcc.sh r05 16 32 64 > gpurun_out/r05_tcc.log 2>&1
out=gpurun_out/r05_run5.txt; : > $out
for lib in product tapA1; do
  if [ $lib = product ]; then unset WN_DEV_LIB; else export WN_DEV_LIB=tools/variants/libwn_$lib.so; fi
  for mode in 0 2 3; do
    echo "== $lib WN_V3_MODE=$mode" >> $out
    for s in 64 96; do WN_TESTING=1 WN_V3_MODE=$mode timeout 300 python tools/rate.py cfg3 $s 3000 2 2>&1 | grep "samples/s" >> $out; done
  done
done
unset WN_DEV_LIB
cat gpurun_out/r05_pytest_gpu_final.txt; cat $out; head -c 1500 gpurun_out/r05_bench_2.json; tail -5 gpurun_out/r05_bench_2.err
