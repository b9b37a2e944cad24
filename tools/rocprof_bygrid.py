"""rocprofv3 kernel trace grouped by (kernel, grid): calls, avg / min duration.  python tools/rocprof_bygrid.py db pattern"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else ""
rows = db.execute("select name, grid_x, grid_y, grid_z, workgroup_x, start, end from kernels").fetchall()
agg = {}
for n, gx, gy, gz, wx, s, e in rows:
    if pat not in n:
        continue
    m = re.search(r"(\w+_kernel)", n)
    k = ((m.group(1) if m else n[:40]) + ("<bf16>" if "DF16b" in n else ""), gx // max(wx, 1), gy, gz)
    a = agg.setdefault(k, [0, 0, 1 << 60])
    a[0] += 1; a[1] += e - s; a[2] = min(a[2], e - s)
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-44s blocks=(%d,%d,%d) calls=%d avg=%.2fus min=%.2fus total=%.3fms" % (k[0], k[1], k[2], k[3], a[0], a[1] / a[0] / 1e3, a[2] / 1e3, a[1] / 1e6))
