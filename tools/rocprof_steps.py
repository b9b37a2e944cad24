"""Steady-state view of a bench.py kernel trace (rocpd SQLite): steps are delimited by copy_multi_kernel (the batch
staging launch that opens every replayed step; adam_kernel for eager runs); prints per-step wall, union-busy and idle time and the idle gaps by the
kernel that follows them.   python tools/rocprof_steps.py bench_results.db [first_step last_step]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
qcol = "stream_id" if "stream_id" in cols else "queue_id"
rows = db.execute("select name, start, end, %s from kernels order by start" % qcol).fetchall()
marks = [i for i, r in enumerate(rows) if "copy_multi_kernel" in r[0]]
if len(marks) < 8:       # eager steps do not stage the batch: the fused Adam launch closes each step
    marks = [i + 1 for i, r in enumerate(rows) if "adam_kernel" in r[0]]
a = int(sys.argv[2]) if len(sys.argv) > 2 else len(marks) // 2
b = int(sys.argv[3]) if len(sys.argv) > 3 else len(marks) - 3
sel = rows[marks[a]:marks[b]]
nsteps = b - a
wall = sel[-1][2] - sel[0][1]
busy, ce = 0, sel[0][1]
gaps = {}
for n, s, e, _ in sel:
    if s > ce:
        m = re.search(r"(\w+_kernel|\w+Functor\w*)", n)
        k = m.group(1) if m else n[:40]
        g = gaps.setdefault(k, [0, 0])
        g[0] += 1; g[1] += s - ce
        busy += 0
    if e > ce:
        busy += e - max(s, ce)
        ce = e
tot = sum(e - s for _, s, e, _ in sel)
print("steps %d..%d: %.3f ms/step wall, %.3f ms busy, %.3f ms idle (%.1f %%), kernel time %.3f ms (overlap %.2f), %d launches/step"
      % (a, b, wall / nsteps / 1e6, busy / nsteps / 1e6, (wall - busy) / nsteps / 1e6, 100.0 * (wall - busy) / wall,
         tot / nsteps / 1e6, tot / busy, len(sel) // nsteps))
print("idle time by the kernel that ends the gap (per step):")
for k, (c, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:14]:
    print("  %-40s %5.1f gaps  %7.1f us" % (k, c / nsteps, t / nsteps / 1e3))

per = {}
for n, s, e, q in sel:
    d = per.setdefault(q, {})
    m = re.search(r"(\w+_kernel)", n)
    k = m.group(1) if m else n[:40]
    a = d.setdefault(k, [0, 0])
    a[0] += 1; a[1] += e - s
print("per stream (kernel time per step):")
for q, d in sorted(per.items(), key=lambda kv: -sum(v[1] for v in kv[1].values())):
    t = sum(v[1] for v in d.values()); c = sum(v[0] for v in d.values())
    top = sorted(d.items(), key=lambda kv: -kv[1][1])[:6]
    print("  stream %s: %.3f ms, %d launches: %s" % (q, t / nsteps / 1e6, c // nsteps,
          ", ".join("%s %.0fus" % (k.replace("_kernel", ""), v[1] / nsteps / 1e3) for k, v in top)))
