#!/bin/bash
# dev tool: build timing-ablation variants of the engine (WN_ABL=n, numerically wrong on purpose) and print the
# chain anatomy of each.  Run on the GPU box:  bash tools/ablate.sh cfg3 "0 1 2 3 5"
cfg=${1:-cfg3}; list=${2:-"0 1 2 3 5"}
for n in $list; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DWN_EXPERIMENT -DWN_ABL=$n -o /tmp/libwn_abl$n.so pytorch-wavenet_amd/csrc/wn_runtime.hip 2>/dev/null || exit 1
  echo "=== WN_ABL=$n"
  WN_DEV_LIB=/tmp/libwn_abl$n.so python tools/profile_chain.py $cfg 1 2>&1 | grep -v "by layer\|^ [0-9]\|amdgpu.ids"
done
