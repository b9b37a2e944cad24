"""per-kernel resources from a rocprofv3 rocpd db: VGPR/AGPR/SGPR, LDS, scratch, typical grid"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, vgpr_count, accum_vgpr_count, sgpr_count, lds_size, scratch_size, grid_x, grid_y, grid_z, workgroup_x, end-start from kernels").fetchall()
agg = {}
for n, v, a, s, l, sc, gx, gy, gz, wx, d in rows:
    m = re.search(r"(\w+_kernel)(I[^E]*E)?", n)
    k = (m.group(0)[:60] if m else n[:60])
    e = agg.setdefault(k, [0, 0, v, a, s, l, sc, wx])
    e[0] += 1; e[1] += d
print("%-62s %6s %9s %5s %5s %5s %7s %7s %4s" % ("kernel", "calls", "total_ms", "vgpr", "agpr", "sgpr", "lds", "scratch", "wg"))
for k, e in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print("%-62s %6d %9.3f %5d %5d %5d %7d %7d %4d" % (k, e[0], e[1] / 1e6, e[2], e[3], e[4], e[5], e[6], e[7]))
