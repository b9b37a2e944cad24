"""Summarise a rocprofv3 (rocpd SQLite) kernel trace: per-kernel calls, total / average duration, share.
    python tools/rocprof_summary.py gpurun_out/prof/bench_results.db [steps] > profiles/<name>.md"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([\w:]+)<(.*)>\(", name)
    if m:
        return "%s<%s>" % (m.group(1), m.group(2)[:48])
    return name.split("(")[0][:90]


def main():
    db = sqlite3.connect(sys.argv[1])
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else None
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else cols[0]
    rows = cur.execute("select %s, start, end from kernels" % namecol).fetchall()
    agg = {}
    for n, s, e in rows:
        a = agg.setdefault(short(n), [0, 0])
        a[0] += 1
        a[1] += e - s
    total = sum(a[1] for a in agg.values())
    print("| kernel | calls | total ms | avg us | % |")
    print("|---|---|---|---|---|")
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("| %s | %d | %.3f | %.2f | %.1f |" % (k, c, t / 1e6, t / c / 1e3, 100.0 * t / total))
    print("\ntotal kernel time %.3f ms over %d dispatches" % (total / 1e6, len(rows)))
    if steps:
        print("per step (%d steps incl. warm-up): %.3f ms kernel time, %d dispatches" % (steps, total / 1e6 / steps, len(rows) // steps))


if __name__ == "__main__":
    main()
