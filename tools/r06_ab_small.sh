#!/bin/bash
# Round-6 A/B for the small single-stream shapes: the round-4 and round-5 trees (their own rate.py + library) against variants of the current source.
mkdir -p gpurun_out; cd /root/repo
run() { # label, tree-or-empty, lib-or-empty
  for c in "cfg2 1" "cfg1 1"; do
    if [ -n "$2" ]; then ( cd $2 && timeout 200 python tools/rate.py $c 16000 3 2>&1 | grep "samples/s" | sed "s/^/$1: /" )
    else WN_DEV_LIB=$3 timeout 200 python tools/rate.py $c 16000 3 2>&1 | grep "samples/s" | sed "s/^/$1: /"; fi
  done
}
for pass in 1 2 3; do
  run r04 tools/variants/r04tree ""
  run r05 tools/variants/r05tree ""
  run product "" pytorch-wavenet_amd/mi355_wavenet/libwn_mi355.so
  run safe_noalign "" tools/variants/libwn_safe_noalign.so
  run fast_align "" tools/variants/libwn_fast_align.so
  run fast_noalign "" tools/variants/libwn_base.so
done 2>&1 | tee gpurun_out/ab_small.txt
