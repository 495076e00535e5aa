#!/bin/bash
# PMC passes on the un-instrumented 64-stream job (variant 3): where do the hand-off accesses go?
mkdir -p gpurun_out
ROOT=/root/repo
cd /tmp && export TMPDIR=/tmp
out=$ROOT/gpurun_out/r02_v3_pmc.txt
: > $out
true
echo >> $out
CMD="python $ROOT/tools/rate.py cfg3 64 2000 1"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum" "TCC_READ_sum TCC_WRITE_sum TCC_ATOMIC_sum" "TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_32B_sum TCC_WRITEBACK_sum" "TCC_NC_REQ_sum TCC_UC_REQ_sum TCC_CC_REQ_sum TCC_RW_REQ_sum"; do
  i=$((i+1))
  rm -rf /tmp/prof_$i
  timeout 200 rocprofv3 --kernel-trace --pmc $set -d /tmp/prof_$i -o p -- $CMD > /tmp/p$i.log 2>&1
  echo "### pass $i: $set" >> $out
  grep -h "samples/s" /tmp/p$i.log | head -1 >> $out
  grep -i "error\|invalid\|not found" /tmp/p$i.log | head -3 >> $out
  python $ROOT/tools/rocprof_summary.py $(find /tmp/prof_$i -name "*.db") 2>&1 | grep "wn_generate.* n=" | cut -c60- >> $out
done
cat $out
