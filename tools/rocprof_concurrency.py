"""Where does a replayed step run ONE kernel at a time?  Walks a mid-run step of a rocprofv3 rocpd db (steps delimited
by copy_multi_kernel) and prints every interval in which a single kernel is in flight, merged per kernel name, plus
the concurrency histogram.   python tools/rocprof_concurrency.py bench_results.db"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
marks = [i for i, r in enumerate(rows) if "copy_multi_kernel" in r[0]]
a = len(marks) * 2 // 3
nsteps = min(10, len(marks) - a - 2)
sel = rows[marks[a]:marks[a + nsteps]]
ev = []
for n, s, e in sel:
    m = re.search(r"(\w+_kernel|\w+Functor)", n)
    k = m.group(1) if m else n[:40]
    ev.append((s, 1, k)); ev.append((e, -1, k))
ev.sort()
live = {}
hist = {}
solo = {}
prev = ev[0][0]
for t, d, k in ev:
    c = sum(live.values())
    hist[c] = hist.get(c, 0) + (t - prev)
    if c == 1:
        name = next(x for x, v in live.items() if v > 0)
        solo[name] = solo.get(name, 0) + (t - prev)
    prev = t
    live[k] = live.get(k, 0) + d
tot = sum(hist.values())
print("per step: %.3f ms;  kernels in flight -> share of time: %s" % (
    tot / nsteps / 1e6, ", ".join("%d: %.1f%%" % (c, 100.0 * v / tot) for c, v in sorted(hist.items()))))
print("time with exactly one kernel in flight, by kernel (us per step):")
for k, v in sorted(solo.items(), key=lambda kv: -kv[1])[:24]:
    print("  %-40s %7.1f" % (k, v / nsteps / 1e3))
