#!/bin/bash
# GPU box: L2 (TCC) counter passes of the generation kernel at 16 / 32 / 64 streams of cfg3 -- hits, misses, reads / writes towards the fabric and their stalls.
# Counters in their own passes with --kernel-trace only (MI355X guide); text summaries into gpurun_out/tcc_<tag>.txt
#   tools/collect_tcc.sh [tag] [streams...]
set -u
TAG=${1:-latest}; shift || true
STREAMS=${*:-"16 32 64"}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/tcc_$TAG.txt
mkdir -p "$ROOT/gpurun_out"; : > "$OUT"
cd /tmp && export TMPDIR=/tmp
for s in $STREAMS; do
  for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_STALL_sum" "TCC_TAG_STALL_sum TCC_WRITEBACK_sum TCC_EA0_RDREQ_32B_sum"; do
    rm -rf /tmp/prof_tcc
    timeout 300 rocprofv3 --kernel-trace --pmc $set -d /tmp/prof_tcc -o t -- python $ROOT/tools/rate.py cfg3 $s 2000 1 > /tmp/tcc.log 2>&1
    echo "### cfg3 x$s: $set" >> "$OUT"
    grep -h 'samples/s' /tmp/tcc.log | head -1 >> "$OUT"
    python $ROOT/tools/rocprof_summary.py $(find /tmp/prof_tcc -name "*.db" | head -1) 2>/dev/null | grep "wn_generate_kernel" | grep -v "^void.*calls" >> "$OUT"
  done
done
cat "$OUT" | head -80
