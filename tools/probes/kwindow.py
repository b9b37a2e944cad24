"""kernels of ONE replayed step of a rocprofv3 kernel trace, in start order, with gaps: usage kwindow.py trace.csv [step_index] [from_ms to_ms]
(times relative to the step's first kernel; the step is delimited by the adam_kernel launches)"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
k = int(sys.argv[2]) if len(sys.argv) > 2 else 10
lo = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
hi = float(sys.argv[4]) if len(sys.argv) > 4 else 1e9
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
a, b = adam[k] + 1, adam[k + 1] + 1
t0 = int(rows[a]["Start_Timestamp"])
last_end = t0
for r in rows[a:b]:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    if s / 1e3 < lo or s / 1e3 > hi:
        continue
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    if n.startswith("_ZN12_GLOBAL__N_1"):
        n = n[17:].lstrip("0123456789")
    wg = int(r["Workgroup_Size_X"]); gr = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // (wg * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"]))
    print("%9.1f  +%6.1f us  q%-3s %-58s %6d x %d" % (s, e - s, r.get("Queue_Id", "?")[-3:], n[:58], gr, wg))
