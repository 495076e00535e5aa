#!/bin/bash
# Runs on the GPU box: tools/traffic_probe (known byte counts, the hand-offs' access patterns) under the two PMC passes; prints counter / known bytes
# per dispatch.   tools/pmc_calibration.sh > gpurun_out/pmc_calibration.txt
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/cal_w /tmp/cal_f
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/cal_w -o w -- $ROOT/tools/traffic_probe > /tmp/cal_w.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/cal_f -o f -- $ROOT/tools/traffic_probe > /tmp/cal_f.log 2>&1
grep -E "^[0-9] k_|^every" /tmp/cal_w.log
python3 - <<PY
import glob, sqlite3
known_kib = 1024 * 1024
for name, pat in (("WRITE_SIZE", "/tmp/cal_w/**/*.db"), ("FETCH_SIZE", "/tmp/cal_f/**/*.db")):
    db = glob.glob(pat, recursive=True)[0]
    con = sqlite3.connect(db)
    rows = con.execute("select dispatch_id, kernel_name, value, duration from counters_collection where counter_name = ? and kernel_name like 'void k_%' or kernel_name like 'k_%' and counter_name = ? order by dispatch_id", (name, name)).fetchall()
    print("# %s (KiB per dispatch) against the %d KiB every kernel moves:" % (name, known_kib))
    for i, (d, k, v, dur) in enumerate(rows):
        print("#   dispatch %d  %-40s %14.1f KiB = %.3f x   (%.0f us)" % (i + 1, k[:40], v, v / known_kib, dur / 1e3))
PY
