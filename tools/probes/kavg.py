"""average duration per kernel name from a rocprofv3 kernel trace CSV (replayed steps only: skips the first third)"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 3: - len(rows) // 6]
acc = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    if n.startswith("_ZN12_GLOBAL__N_1"):
        n = n[17:].lstrip("0123456789")
    acc[n[:46]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
pat = sys.argv[2:] 
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    if not pat or any(p in k for p in pat):
        print("%-48s n=%5d avg %7.2f us  total %8.2f ms" % (k, len(v), sum(v) / len(v), sum(v) / 1e3))
