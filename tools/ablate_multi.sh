#!/bin/bash
# dev tool: timing ablations of the multi-stream kernel (results numerically wrong when WN_ABL != 0)
for n in ${1:-"0 7"}; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DWN_EXPERIMENT -DWN_ABL=$n -o /tmp/libwn_abl$n.so pytorch-wavenet_amd/csrc/wn_runtime.hip 2>/dev/null || exit 1
  echo "=== WN_ABL=$n"
  WN_DEV_LIB=/tmp/libwn_abl$n.so python tools/profile_chain.py cfg3 64 2>&1 | grep "loop period\|layer 25\|published"
done
