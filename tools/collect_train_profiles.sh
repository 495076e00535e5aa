#!/bin/bash
# Runs on the GPU box (gpurun): rocprofv3 kernel stats and the two PMC passes for the config-5 training step (bf16 operands, and fp32 with
# "fp32" as the second argument); text summaries land in gpurun_out/ (copy the ones to keep into profiles/), pmc_traffic_train.json next to them.
#   tools/collect_train_profiles.sh [tag] [bf16|fp32]
set -u
TAG=${1:-latest}
PREC=${2:-bf16}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
FLAG="--only-bf16"; [ "$PREC" = "fp32" ] && FLAG="--only-fp32"
CMD="python $ROOT/tools/bench_train.py 32 16000 --no-torch $FLAG"
rm -rf /tmp/prof_tkt /tmp/prof_tf /tmp/prof_tw
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_tkt -o kt -- $CMD > /tmp/tkt.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/prof_tf -o f -- $CMD > /tmp/tf.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/prof_tw -o w -- $CMD > /tmp/tw.log 2>&1
cd "$ROOT"
{
  echo "# rocprofv3 (ROCm 7.2) summaries of: $CMD"
  echo "# pass 1: --kernel-trace --stats; pass 2: --pmc FETCH_SIZE; pass 3: --pmc WRITE_SIZE (counters in separate passes)"
  grep -h 'ms / step' /tmp/tkt.log | head -2
  python tools/rocprof_summary.py $(find /tmp/prof_tkt -name "*.db" | sort) 2>&1 | head -40
} > "$OUT/rocprofv3_train_${PREC}_$TAG.txt" 2>&1
python tools/make_pmc_train_json.py $(find /tmp/prof_tf -name "*.db" | head -1) $(find /tmp/prof_tw -name "*.db" | head -1) "$PREC" "profiles/${TAG}_rocprofv3_train_step_cfg5_${PREC}.txt" "$OUT/pmc_traffic_train.json" $(find /tmp/prof_tkt -name "*.db" | head -1) >> "$OUT/rocprofv3_train_${PREC}_$TAG.txt" 2>&1
head -c 2500 "$OUT/rocprofv3_train_${PREC}_$TAG.txt"; tail -5 "$OUT/rocprofv3_train_${PREC}_$TAG.txt"
