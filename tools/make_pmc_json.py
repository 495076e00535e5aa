"""Runs on the GPU box after tools/collect_profiles.sh: turns the two rocprofv3 PMC databases (FETCH_SIZE pass, WRITE_SIZE pass) of the
default bench workload into the pmc_traffic.json that bench.py reads -- with the kernel, its form (wn_get_info) and the date, so that
bench.py can refuse the figure when another kernel or form runs.   python tools/make_pmc_json.py <fetch.db> <write.db> <summary file> <out.json>"""
import datetime
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-wavenet_amd"))
from mi355_wavenet import engine, synth  # noqa: E402


def counter(db, name):
    con = sqlite3.connect(db)
    rows = con.execute("select kernel_name, avg(value), count(*) from counters_collection where counter_name = ? and kernel_name like '%wn_generate_kernel%' "
                       "group by kernel_name", (name,)).fetchall()
    assert len(rows) == 1, rows
    return rows[0]


def main():
    fdb, wdb, summary, out = sys.argv[1:5]
    streams, samples = 64, 2000
    kname, fetch, n1 = counter(fdb, "FETCH_SIZE")
    _, write, n2 = counter(wdb, "WRITE_SIZE")
    cfg = synth.CONFIGS["cfg3"]
    eng = engine.Engine(cfg, synth.init_weights(cfg, seed=0), n_streams=streams)
    info = eng.info()
    eng.close()
    doc = {
        "_comment": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, KiB per dispatch, averages over %d / %d dispatches) on MI355X for one "
                    "64-stream job of %d timesteps (tools/collect_profiles.sh -> tools/make_pmc_json.py); raw counter values; traffic = fetch_correction x fetch_kib + write_kib "
                    "(calibrated on the hand-offs' own access patterns: profiles/r04_pmc_calibration.txt)" % (n1, n2, samples),
        "cfg3x64": {
            "streams": streams, "samples_per_launch": samples, "fetch_kib": round(fetch, 1), "write_kib": round(write, 1), "kernels_per_job": 1,
            "kernel": kname, "date": datetime.date.today().isoformat(), "summary": summary,
            "fetch_correction": 2.0,
            "calibration": "profiles/r04_pmc_calibration.txt: WRITE_SIZE exact, FETCH_SIZE reports half of the missed bytes for 8- and 16-byte sc1 loads",
            "form": {k: info[k] for k in ("kernel_variant", "streams_per_item", "head_replicas", "n_samplers", "n_workgroups")},
        },
    }
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps(doc["cfg3x64"]))


if __name__ == "__main__":
    main()
