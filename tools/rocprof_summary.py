"""Turns rocprofv3's results.db (ROCm 7.2 writes SQLite by default) into the small text summaries committed under
profiles/:  python tools/rocprof_summary.py <results.db> [<results.db> ...]"""
import sqlite3
import sys


def main():
    for db in sys.argv[1:]:
        con = sqlite3.connect(db)
        print("## %s" % db)
        try:
            rows = con.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
            print("kernel stats (durations in us):")
            print("%-70s %6s %16s %14s %8s" % ("name", "calls", "total_us", "avg_us", "pct"))
            for r in rows:
                print("%-70s %6d %16.1f %14.1f %8.3f" % (r[0][:70], r[1], r[2], r[3], r[4]))
        except sqlite3.Error as e:
            print("no kernel stats:", e)
        try:
            rows = con.execute("select kernel_name, counter_name, count(*), avg(value), min(value), max(value), avg(duration) "
                               "from counters_collection group by kernel_name, counter_name").fetchall()
            if rows:
                print("PMC counters per dispatch (FETCH_SIZE / WRITE_SIZE are in KiB; gfx950 note in MI355X_MICROARCH.md: "
                      "FETCH_SIZE under-counts wide coalesced reads by 2x, other patterns uncalibrated):")
                for r in rows:
                    print("%-60s %-12s n=%d avg=%.1f min=%.1f max=%.1f avg_dispatch_ns=%.0f" % (r[0][:60], r[1], r[2], r[3], r[4], r[5], r[6]))
        except sqlite3.Error as e:
            print("no counters:", e)
        print()


if __name__ == "__main__":
    main()
