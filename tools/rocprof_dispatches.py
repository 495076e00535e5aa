"""Per-dispatch listing (name, grid, duration) from a rocprofv3 results.db: python tools/rocprof_dispatches.py <db> [max_rows]"""
import sqlite3
import sys


def main():
    con = sqlite3.connect(sys.argv[1])
    limit = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
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
    print("# " + " ".join(want))
    for r in rows:
        print(" ".join(str(x)[:60] if i == 0 else str(x) for i, x in enumerate(r)))


if __name__ == "__main__":
    main()
