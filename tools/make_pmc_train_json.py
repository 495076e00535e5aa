"""Runs on the GPU box after tools/collect_train_profiles.sh: sums FETCH_SIZE / WRITE_SIZE (rocprofv3 PMC passes, KiB per dispatch) over EVERY kernel of the
profiled training run and divides by the number of steps (= dispatches of wn_fwd_start, one per forward): HBM bytes per training step, into the
pmc_traffic_train.json that bench.py replays next to the algorithmic figure.
    python tools/make_pmc_train_json.py <fetch.db> <write.db> <bf16|fp32> <summary file> <out.json> [<stats.db>]
With the --kernel-trace --stats database of the same command as a sixth argument: per kernel, counted bytes (2 x FETCH + WRITE) over its own time in the
UNPERTURBED pass = the rate it moves data at, next to the chip's measured streaming ceilings (profiles/r04_pmc_calibration.txt: 4.45 TB/s load, 5.56 TB/s store)."""
import datetime
import json
import os
import sqlite3
import sys


def total(db, name):
    con = sqlite3.connect(db)
    tot, n = con.execute("select sum(value), count(*) from counters_collection where counter_name = ?", (name,)).fetchone()
    steps = con.execute("select count(*) from counters_collection where counter_name = ? and kernel_name like '%wn_fwd_start%'", (name,)).fetchone()[0]
    top = con.execute("select kernel_name, sum(value), count(*) from counters_collection where counter_name = ? group by kernel_name order by sum(value) desc limit 8",
                      (name,)).fetchall()
    return float(tot), int(n), int(steps), top


def main():
    fdb, wdb, prec, summary, out = sys.argv[1:6]
    f, nf, sf, topf = total(fdb, "FETCH_SIZE")
    w, nw, sw, topw = total(wdb, "WRITE_SIZE")
    assert sf > 0 and sw > 0, (sf, sw)
    doc = json.load(open(out)) if os.path.exists(out) else {}
    doc["_comment"] = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, KiB per dispatch) summed over every kernel of tools/bench_train.py 32 16000 "
                       "and divided by the steps of the run (dispatches of wn_fwd_start); raw counter values; traffic = fetch_correction x fetch_kib + write_kib "
                       "(MI355X_MICROARCH.md and profiles/r04_pmc_calibration.txt: FETCH_SIZE reports half of the bytes of a coalesced streaming read on gfx950)")
    doc["train5_%s" % prec] = {"clips": 32, "clip_samples": 16000, "steps_profiled": sf, "dispatches": nf, "fetch_kib": round(f / sf, 1), "write_kib": round(w / sw, 1),
                               "date": datetime.date.today().isoformat(), "summary": summary, "fetch_correction": 2.0,
                               "calibration": "profiles/r04_pmc_calibration.txt (and MI355X_MICROARCH.md: FETCH_SIZE reports half of a coalesced streaming read)"}
    json.dump(doc, open(out, "w"), indent=1)
    print("# HBM bytes per step from the PMC passes (%s): 2 x FETCH_SIZE %.1f GB + WRITE_SIZE %.1f GB = %.1f GB over %d dispatches / %d steps (FETCH_SIZE counts half: profiles/r04_pmc_calibration.txt)" % (
        prec, 2 * f / sf * 1024 / 1e9, w / sw * 1024 / 1e9, (2 * f / sf + w / sw) * 1024 / 1e9, nf, sf))
    for label, top, steps in (("FETCH_SIZE", topf, sf), ("WRITE_SIZE", topw, sw)):
        print("# largest by %s (GB per step):" % label)
        for k, v, c in top:
            print("#   %-90s %8.2f GB  (%d dispatches)" % (k[:90], v / steps * 1024 / 1e9, c))
    if len(sys.argv) > 6:
        per = {}
        for db, name, mult in ((fdb, "FETCH_SIZE", 2.0), (wdb, "WRITE_SIZE", 1.0)):
            for k, v in sqlite3.connect(db).execute("select kernel_name, sum(value) from counters_collection where counter_name = ? group by kernel_name", (name,)):
                per[k] = per.get(k, 0.0) + mult * float(v) * 1024
        rows = sqlite3.connect(sys.argv[6]).execute("select name, total_calls, total_duration from top_kernels").fetchall()
        print("# counted bytes (2 x FETCH_SIZE + WRITE_SIZE) over the kernel's own time in the --stats pass; ceilings of this chip: 4.45 TB/s streaming load, 5.56 TB/s streaming")
        print("# store (profiles/r04_pmc_calibration.txt); kernels overlap on two streams, so the rates of concurrent kernels ADD UP to what the memory system delivers:")
        for name, calls, total_us in rows[:10]:   # (top_kernels.total_duration is in microseconds: tools/rocprof_summary.py prints the same column)
            b = per.get(name)
            if b is None:
                cands = [k for k in per if k.startswith(name[:40])]
                b = per[cands[0]] if len(cands) == 1 else None
            if b is not None and total_us > 0:
                print("#   %-78s %7.2f GB/step in %6.2f ms/step = %5.2f TB/s" % (name[:78], b / sf / 1e9, total_us / sf / 1e3, b / (total_us * 1e-6) / 1e12))


if __name__ == "__main__":
    main()
