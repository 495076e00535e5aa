#!/bin/bash
# round 5, run 4: code alignment of branch targets / poll loops (x1 is 4.5 % sensitive to a 4-byte shift of the code), pending tests
mkdir -p gpurun_out
out=gpurun_out/r05_run4.txt; : > $out
V=tools/variants
for rep in 1 2; do
for lib in product nop5 al6 al5 sp6 al6sp6 al6safe; do
  if [ $lib = product ]; then unset WN_DEV_LIB; else export WN_DEV_LIB=$V/libwn_$lib.so; fi
  echo "== $lib (rep $rep)" >> $out
  for s in 1 64; do timeout 300 python tools/rate.py cfg3 $s 3000 2 2>&1 | grep "samples/s" >> $out; done
done
done
unset WN_DEV_LIB
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_training.py -m gpu -q -s -k "resident or bf16_step or finds_the_cus" 2>&1 | grep -v Warning | tail -60 > gpurun_out/r05_pytest_gpu_3.txt
cat $out; cat gpurun_out/r05_pytest_gpu_3.txt
