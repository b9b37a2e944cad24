"""One replayed step of a rocprofv3 rocpd kernel trace as a table: start offset, duration, queue, kernel — every launch of a
mid-run step (steps delimited by copy_multi_kernel).   python tools/rocprof_gantt.py bench_results.db [from_us [to_us]]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else "stream_id"
rows = db.execute("select name, start, end, %s from kernels order by start" % qcol).fetchall()
marks = [i for i, r in enumerate(rows) if "copy_multi_kernel" in r[0]]
a = len(marks) * 2 // 3
sel = rows[marks[a]:marks[a + 1]]
t0 = sel[0][1]
lo = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
hi = float(sys.argv[3]) if len(sys.argv) > 3 else 1e9
qs = sorted({r[3] for r in sel})
print("step %d: %d launches, %.3f ms; queues %s" % (a, len(sel), (max(r[2] for r in sel) - t0) / 1e6, qs))
for n, s, e, q in sel:
    if lo <= (s - t0) / 1e3 <= hi:
        m = re.search(r"(\w+_kernel|\w+Functor)", n)
        print("%8.1f %7.1f  %s%s" % ((s - t0) / 1e3, (e - s) / 1e3, "      " * qs.index(q), (m.group(1) if m else n[:30]).replace("_kernel", "")))
