"""kernel sequence of the last step in a rocprofv3 rocpd db, per stream: name, grid, duration, gap to previous"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end, stream_id, queue_id, grid_x, workgroup_x from kernels order by start").fetchall()
# last step = from the last adam_kernel backwards to the previous adam_kernel
idx = [i for i, r in enumerate(rows) if "adam_kernel" in r[0]]
lo, hi = idx[-2] + 1, idx[-1] + 1
step = rows[lo:hi]
print("last step: %d kernels, %.3f ms from first start to last end" % (len(step), (step[-1][2] - step[0][1]) / 1e6))
last_end = {}
for n, s, e, st, q, gx, wx in step:
    m = re.search(r"(\w+_kernel)", n)
    k = m.group(1) if m else n[:50]
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = e
    print("q%d  +%7.1f us  %-34s blocks=%-6d %7.1f us   gap %5.1f" % (q, (s - step[0][1]) / 1e3, k[:34], gx // max(wx, 1), (e - s) / 1e3, gap))
