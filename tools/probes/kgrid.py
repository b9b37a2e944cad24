"""per (kernel, grid, block) statistics of the replayed steps of a rocprofv3 kernel trace CSV: launches per step, average
duration in the step, share of the summed kernel time.  usage: kgrid.py trace.csv steps [pattern...]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2])
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 3: - len(rows) // 6]
acc = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    if n.startswith("_ZN12_GLOBAL__N_1"):
        n = n[17:].lstrip("0123456789")
    wg = int(r["Workgroup_Size_X"]) * int(r.get("Workgroup_Size_Y", 1) or 1) * int(r.get("Workgroup_Size_Z", 1) or 1)
    gr = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)
    acc[(n[:52], gr // max(wg, 1), wg)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in acc.values())
frac = len(rows) / float(sum(1 for _ in rows))
pat = sys.argv[3:]
print("summed kernel time in the window: %.2f ms" % (tot / 1e3))
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    if pat and not any(p in k[0] for p in pat):
        continue
    print("%-54s blocks %6d x %4d  n=%5d  avg %7.2f us  min %7.2f  %5.2f %%" % (k[0], k[1], k[2], len(v), sum(v) / len(v), min(v), 100 * sum(v) / tot))
