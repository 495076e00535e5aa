#!/bin/bash
# Runs on the GPU box (gpurun): rocprofv3 kernel trace of the config-5 training step with bf16 operands (5 steps: 2 warm-up + 3 timed).
#   tools/profile_train.sh [tag]      -> gpurun_out/rocprofv3_train_<tag>.txt
set -u
TAG=${1:-latest}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/tools/bench_train.py 32 16000 --no-torch --only-bf16"
rm -rf /tmp/prof_tr
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_tr -o tr -- $CMD > /tmp/tr.log 2>&1
if [ "${PMC:-0}" = "1" ]; then   # HBM traffic per kernel: counters in their own passes (no --stats, no other trace domains)
  rm -rf /tmp/prof_trf /tmp/prof_trw
  timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/prof_trf -o f -- $CMD > /tmp/trf.log 2>&1
  timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/prof_trw -o w -- $CMD > /tmp/trw.log 2>&1
fi
if [ "${SQ:-0}" = "1" ]; then    # where the waves' cycles go (quad-cycles; WAIT_ANY = parked on s_waitcnt / barrier) and the LDS conflict share
  rm -rf /tmp/prof_trs
  timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS -d /tmp/prof_trs -o s -- $CMD > /tmp/trs.log 2>&1
fi
cd "$ROOT"
DB=$(find /tmp/prof_tr -name "*.db" | head -1)
{
  echo "# rocprofv3 --kernel-trace --stats (ROCm 7.2) of: $CMD"
  grep -h 'ms / step' /tmp/tr.log
  python tools/rocprof_summary.py $DB | head -24
  echo "# the same dispatches by (kernel, grid): which product costs what"
  python tools/rocprof_dispatches.py $DB 100000 40 --group
  echo "# timeline of the last three steps: queue occupancy and the main queue's gaps (tools/rocprof_timeline.py)"
  python tools/rocprof_timeline.py $DB ${WINDOW_MS:-170} 14
  if [ "${PMC:-0}" = "1" ]; then
    echo "# PMC passes (FETCH_SIZE / WRITE_SIZE in KiB per dispatch; raw counter values)"
    python tools/rocprof_summary.py $(find /tmp/prof_trf /tmp/prof_trw -name "*.db" | sort) | grep -v "^kernel stats\|^name \|^void at::\|^__amd\|^## " | grep "n=" | grep "wn_"
  fi
  if [ "${SQ:-0}" = "1" ]; then
    echo "# SQ pass (per dispatch averages)"
    python tools/rocprof_summary.py $(find /tmp/prof_trs -name "*.db" | sort) | grep "n=" | grep "wn_"
    tail -3 /tmp/trs.log
  fi
} > "$OUT/rocprofv3_train_$TAG.txt" 2>&1
head -c 6000 "$OUT/rocprofv3_train_$TAG.txt"
