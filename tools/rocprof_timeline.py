"""Timeline view of a rocprofv3 (rocpd SQLite) kernel trace: per-queue/stream busy time, union busy time and
idle share over the steady-state window, and the biggest gaps.
    python tools/rocprof_timeline.py bench_results.db [skip_fraction]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    print("columns:", cols)
    qcol = "stream_id" if "stream_id" in cols else ("queue_id" if "queue_id" in cols else None)
    q2 = "queue_id" if "queue_id" in cols else qcol
    rows = cur.execute("select name, start, end, %s, %s from kernels order by start" % (qcol, q2)).fetchall()
    t0, t1 = rows[0][1], rows[-1][2]
    lo = t0 + (t1 - t0) * skip
    rows = [r for r in rows if r[1] >= lo]
    wall = rows[-1][2] - rows[0][1]
    per = {}
    for n, s, e, q, qq in rows:
        a = per.setdefault((q, qq), [0, 0])
        a[0] += 1
        a[1] += e - s
    print("window %.3f ms, %d dispatches" % (wall / 1e6, len(rows)))
    for k, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        print("  stream/queue %s: %d kernels, busy %.3f ms (%.1f %% of window)" % (k, c, t / 1e6, 100.0 * t / wall))
    # union of busy intervals
    iv = sorted((s, e) for _, s, e, _, _ in rows)
    busy, cs, ce = 0, iv[0][0], iv[0][1]
    gaps = []
    for s, e in iv[1:]:
        if s > ce:
            busy += ce - cs
            gaps.append((s - ce, ce))
            cs, ce = s, e
        else:
            ce = max(ce, e)
    busy += ce - cs
    print("union busy %.3f ms = %.1f %% of window; sum of kernel time %.3f ms (overlap factor %.2f)" % (
        busy / 1e6, 100.0 * busy / wall, sum(e - s for s, e in iv) / 1e6, sum(e - s for s, e in iv) / busy))
    gaps.sort(reverse=True)
    print("gaps: n=%d total %.3f ms; >5us: %d (%.3f ms); top: %s" % (
        len(gaps), sum(g for g, _ in gaps) / 1e6, sum(1 for g, _ in gaps if g > 5000),
        sum(g for g, _ in gaps if g > 5000) / 1e6, [round(g / 1e3, 1) for g, _ in gaps[:12]]))


if __name__ == "__main__":
    main()
