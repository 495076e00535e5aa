#!/bin/bash
# Round-6 A/B of engine variants built by tools/build_variant.py:   tools/r06_ab.sh <tag> "<variant names>" [cfg] [streams] [samples] [reps]
TAG=$1; LIST=$2; CFG=${3:-cfg3}; NS=${4:-64}; N=${5:-3000}; REPS=${6:-3}
mkdir -p gpurun_out
cd /root/repo
for pass in 1 2; do
  for v in $LIST; do
    echo "=== $v (pass $pass)"
    for s in $NS; do WN_DEV_LIB=tools/variants/libwn_$v.so timeout 300 python tools/rate.py $CFG $s $N $REPS 2>&1 | grep "samples/s"; done
  done
done 2>&1 | tee gpurun_out/ab_$TAG.txt
