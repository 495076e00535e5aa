#!/bin/bash
# dev tool: build timing-ablation variants of the engine (-DWN_EXPERIMENT -DWN_V3_ABL=n: 1 no skip-group work, 2 no queue-group work, 3 neither;
# numerically wrong on purpose) and print their rates.  Run on the GPU box:  bash tools/ablate.sh cfg3 "64 128" "1 2 3"
cfg=${1:-cfg3}; streams=${2:-"64"}; list=${3:-"1 2 3"}
mkdir -p tools/variants
for n in $list; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DWN_EXPERIMENT -DWN_V3_ABL=$n -o tools/variants/libwn_abl$n.so pytorch-wavenet_amd/csrc/wn_runtime.hip pytorch-wavenet_amd/csrc/wn_stacked.hip 2>/dev/null || exit 1
  echo "=== WN_V3_ABL=$n"
  for s in $streams; do WN_DEV_LIB=tools/variants/libwn_abl$n.so python tools/rate.py $cfg $s 3000 2 2>&1 | grep "samples/s"; done
done
