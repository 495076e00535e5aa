#!/bin/bash
# Compile-time resource usage of every kernel in libwn_mi355.so (no GPU needed): VGPRs, AGPRs, scratch, occupancy, LDS.
#   tools/kernel_resources.sh > profiles/archive/r01_kernel_resources.txt
ROOT=$(cd "$(dirname "$0")/.." && pwd)
echo "# hipcc --offload-arch=gfx950 -O3 -Rpass-analysis=kernel-resource-usage on pytorch-wavenet_amd/csrc/wn_runtime.hip + wn_stacked.hip"
echo "# name | VGPRs | AGPRs | scratch bytes/lane | occupancy waves/SIMD | static LDS bytes/block (dynamic LDS is set at launch)"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Rpass-analysis=kernel-resource-usage -I"$ROOT/include" -o /tmp/wn_res_check.so \
      "$ROOT/pytorch-wavenet_amd/csrc/wn_runtime.hip" "$ROOT/pytorch-wavenet_amd/csrc/wn_stacked.hip" 2>&1 \
  | grep -E "Function Name|VGPRs:|AGPRs|ScratchSize|Occupancy|LDS Size" | sed 's/.*remark: *//; s/ \[-Rpass.*//' | paste - - - - - - \
  | sed 's/Function Name: //; s/VGPRs: //; s/AGPRs: //; s/ScratchSize \[bytes\/lane\]: //; s/Occupancy \[waves\/SIMD\]: //; s/LDS Size \[bytes\/block\]: //' \
  | while IFS=$'\t' read -r name v a s o l; do printf "%-90s | %3s | %3s | %3s | %s | %s\n" "$(echo "$name" | c++filt | cut -c1-90)" "$v" "$a" "$s" "$o" "$l"; done | sort -u
rm -f /tmp/wn_res_check.so
