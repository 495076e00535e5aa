#!/bin/bash
# round 5: the slot re-use form of cfg3's kernel -- tests, rates by stream count with the form pinned off / on, HBM traffic of the 128-stream job both ways
mkdir -p gpurun_out
out=gpurun_out/r05_slots.txt; : > $out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "slot_form or throughput_forms or rounds" 2>&1 | tail -4 >> $out
export WN_TESTING=1
for pin in 0 4; do
  echo "== WN_V3_SLOTS=$pin" >> $out
  for s in 64 80 96 112 128 150; do WN_V3_SLOTS=$pin timeout 300 python tools/rate.py cfg3 $s 2500 2 2>&1 | grep "samples/s" >> $out; done
done
echo "== planner's choice" >> $out
for s in 64 96 128; do timeout 300 python tools/rate.py cfg3 $s 2500 2 2>&1 | grep "samples/s" >> $out; done
cd /tmp && export TMPDIR=/tmp
for pin in 0 4; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/prof_s
    WN_V3_SLOTS=$pin timeout 300 rocprofv3 --kernel-trace --pmc $c -d /tmp/prof_s -o s -- python $GRAFT_REPO_ROOT/tools/rate.py cfg3 128 2000 1 > /tmp/s.log 2>&1
    echo "### cfg3 x128 WN_V3_SLOTS=$pin: $c" >> $GRAFT_REPO_ROOT/$out
    python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $(find /tmp/prof_s -name "*.db" | head -1) 2>/dev/null | grep "wn_generate_kernel" | grep "$c" >> $GRAFT_REPO_ROOT/$out
  done
done
cat $GRAFT_REPO_ROOT/$out
