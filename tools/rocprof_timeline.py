"""Timeline of the LAST `window_ms` of a rocprofv3 --kernel-trace results.db: how busy each HIP queue was, how much of the wall clock no kernel ran
at all, and where the main queue's gaps are (the dispatches in front of which it sat idle longest).

    python tools/rocprof_timeline.py <db> [window_ms=170] [top=12]

Answers "what would a hipGraph / a dataflow kernel buy": the sum of the main queue's gaps is the most a capture can remove."""
import sqlite3
import sys


def main():
    con = sqlite3.connect(sys.argv[1])
    window_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 170.0
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
    qcol = next((c for c in ("queue_id", "stream_id", "queue") if c in cols), None)
    sel = "name, start, end" + (", %s" % qcol if qcol else "")
    rows = con.execute("select %s from kernels order by start" % sel).fetchall()
    if not rows:
        print("no kernel rows; columns:", cols)
        return
    t_end = max(r[2] for r in rows)
    t0 = t_end - window_ms * 1e6
    rows = [r for r in rows if r[1] >= t0]
    print("# columns: %s; queue column: %s; %d dispatches in the last %.0f ms" % (cols, qcol, len(rows), window_ms))
    by_q = {}
    for r in rows:
        by_q.setdefault(r[3] if qcol else 0, []).append(r)
    wall = (t_end - min(r[1] for r in rows)) / 1e6

    def union(iv):
        iv = sorted(iv)
        tot, cs, ce = 0, None, None
        for s, e in iv:
            if cs is None:
                cs, ce = s, e
            elif s <= ce:
                ce = max(ce, e)
            else:
                tot += ce - cs; cs, ce = s, e
        return tot + (ce - cs if cs is not None else 0)

    busy_all = union([(r[1], r[2]) for r in rows]) / 1e6
    print("wall %.2f ms; some kernel running %.2f ms (%.1f %%); nothing running %.2f ms" % (wall, busy_all, 100 * busy_all / wall, wall - busy_all))
    for q, rs in sorted(by_q.items(), key=lambda kv: -sum(r[2] - r[1] for r in kv[1])):
        busy = union([(r[1], r[2]) for r in rs]) / 1e6
        total = sum(r[2] - r[1] for r in rs) / 1e6
        print("queue %s: %5d dispatches, kernel time %.2f ms, busy %.2f ms (%.1f %% of the wall)" % (q, len(rs), total, busy, 100 * busy / wall))
    main_q = max(by_q.items(), key=lambda kv: sum(r[2] - r[1] for r in kv[1]))[0]
    rs = by_q[main_q]
    gaps = []
    for a, b in zip(rs, rs[1:]):
        gaps.append((b[1] - a[2], a[0][:48], b[0][:48]))
    pos = [g for g in gaps if g[0] > 0]
    print("main queue %s: %d gaps > 0, sum %.2f ms, median %.1f us; gaps > 20 us: %d (sum %.2f ms)" % (
        main_q, len(pos), sum(g[0] for g in pos) / 1e6, sorted(g[0] for g in pos)[len(pos) // 2] / 1e3 if pos else 0,
        sum(1 for g in pos if g[0] > 20000), sum(g[0] for g in pos if g[0] > 20000) / 1e6))
    agg = {}
    for g in pos:
        c = agg.setdefault((g[1], g[2]), [0, 0])
        c[0] += 1; c[1] += g[0]
    print("# largest gap classes on the main queue (after -> before): count, total ms, mean us")
    for k, (n, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print("  %-48s -> %-48s %5d %8.2f %8.1f" % (k[0], k[1], n, tot / 1e6, tot / n / 1e3))


if __name__ == "__main__":
    main()
