"""Per-dispatch listing (name, grid, duration) from a rocprofv3 results.db: python tools/rocprof_dispatches.py <db> [max_rows]
   --group: one line per (kernel name, grid) instead -- calls, average and total duration (separates the uses of one GEMM kernel)"""
import sqlite3
import sys


def main():
    group = "--group" in sys.argv
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    con = sqlite3.connect(args[0])
    limit = int(args[1]) if len(args) > 1 else 100000
    names = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
    view = "kernels" if "kernels" in names else None
    if view is None:
        print("views/tables:", names)
        return
    cols = [r[1] for r in con.execute("pragma table_info(%s)" % view)]
    want = [c for c in ("name", "grid_x", "grid_y", "grid_z", "grid_size_x", "grid_size_y", "grid_size_z", "workgroup_x", "workgroup_size_x", "start", "end", "duration", "lds_size", "lds_block_size") if c in cols]
    print("# columns available:", cols)
    order = "start" if "start" in cols else cols[0]
    rows = con.execute("select %s from %s order by %s limit %d" % (", ".join(want), view, order, limit)).fetchall()
    if group:
        gi = [i for i, c in enumerate(want) if c.startswith("grid")]
        di = want.index("duration") if "duration" in want else None
        agg = {}
        for r in rows:
            dur = r[di] if di is not None else r[want.index("end")] - r[want.index("start")]
            k = (str(r[0])[:64],) + tuple(r[i] for i in gi)
            c = agg.setdefault(k, [0, 0.0])
            c[0] += 1; c[1] += dur
        print("# name grid calls avg_us total_ms")
        top = int(args[2]) if len(args) > 2 else 40
        for k, (n, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
            print("%-64s %-22s %6d %10.1f %10.2f" % (k[0], "x".join(str(x) for x in k[1:]), n, tot / n / 1e3, tot / 1e6))
        return
    print("# " + " ".join(want))
    for r in rows:
        print(" ".join(str(x)[:60] if i == 0 else str(x) for i, x in enumerate(r)))


if __name__ == "__main__":
    main()
