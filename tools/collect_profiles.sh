#!/bin/bash
# Runs on the GPU box (gpurun): rocprofv3 kernel stats and the two PMC passes for the default bench workload; the SQLite
# databases stay in /tmp, only text summaries land in gpurun_out/ (copy the ones to keep into profiles/).
#   tools/collect_profiles.sh [tag]
set -u
TAG=${1:-latest}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/tools/rate.py cfg3 64 2000 2"   # one wn_generate job of 2000 timesteps x 64 streams per repetition (the engine-level leg of bench.py)
rm -rf /tmp/prof_kt /tmp/prof_f /tmp/prof_w
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt -- $CMD > /tmp/kt.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/prof_f -o f -- $CMD > /tmp/f.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/prof_w -o w -- $CMD > /tmp/w.log 2>&1
cd "$ROOT"
{
  echo "# rocprofv3 (ROCm 7.2) summaries of: $CMD"
  echo "# pass 1: --kernel-trace --stats; pass 2: --pmc FETCH_SIZE; pass 3: --pmc WRITE_SIZE (counters in separate passes)"
  grep -h 'samples/s' /tmp/kt.log | head -1
  python tools/rocprof_summary.py $(find /tmp/prof_kt /tmp/prof_f /tmp/prof_w -name "*.db" | sort)
} > "$OUT/rocprofv3_$TAG.txt" 2>&1
python tools/make_pmc_json.py $(find /tmp/prof_f -name "*.db" | head -1) $(find /tmp/prof_w -name "*.db" | head -1) "profiles/${TAG}_rocprofv3_cfg3x64.txt" "$OUT/pmc_traffic.json"
head -c 3000 "$OUT/rocprofv3_$TAG.txt"
