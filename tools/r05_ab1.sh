#!/bin/bash
# round 5, A/B 1: when does a late layer's queue group request its tap (behind barrier A instead of behind its dot), deferred skip / queue chunks
out=gpurun_out/r05_ab1.txt; mkdir -p gpurun_out; : > $out
V=tools/variants
for s in 64 7; do WN_DEV_LIB=$V/libwn_tapA.so timeout 300 python tools/quick_check.py cfg3 $s 2>&1 | grep quick_check >> $out; done
for lib in product base_safe tapA tapA_sd7 tapA_sd7_qd7 sd7; do
  echo "== $lib" >> $out
  if [ $lib = product ]; then unset WN_DEV_LIB; else export WN_DEV_LIB=$V/libwn_$lib.so; fi
  timeout 300 python tools/rate.py cfg3 64 3000 2 2>&1 | grep "samples/s" >> $out
done
for lib in product base_safe tapA; do
  echo "== $lib x1 / x128 / x32" >> $out
  if [ $lib = product ]; then unset WN_DEV_LIB; else export WN_DEV_LIB=$V/libwn_$lib.so; fi
  timeout 300 python tools/rate.py cfg3 1 3000 2 2>&1 | grep "samples/s" >> $out
  timeout 300 python tools/rate.py cfg3 128 2000 2 2>&1 | grep "samples/s" >> $out
  timeout 300 python tools/rate.py cfg3 32 3000 2 2>&1 | grep "samples/s" >> $out
done
unset WN_DEV_LIB
echo "== dilation probe (product) x64" >> $out
timeout 600 python tools/dilation_probe.py 64 2>&1 | grep "layers" >> $out
echo "== anatomy tapA x64" >> $out
WN_DEV_LIB=$V/libwn_tapA.so timeout 300 python tools/profile_chain.py cfg3 64 2>&1 | grep -v "^  layer\|^   L" >> $out
echo "== anatomy product x64" >> $out
timeout 300 python tools/profile_chain.py cfg3 64 2>&1 | grep -v "^  layer\|^   L" >> $out
cat $out
