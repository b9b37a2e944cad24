"""Kernel sequence of ONE replayed step from a rocprofv3 --kernel-trace CSV (start offset, duration, queue, blocks, name).
    python tools/step_sequence.py <kernel_trace.csv> [step_index_from_end]"""
import csv
import sys


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    if n.startswith("_ZN12_GLOBAL__N_1"):
        n = n[17:].lstrip("0123456789")
    return n[:60]


rows = []
for r in csv.DictReader(open(sys.argv[1])):
    wg = max(int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1), 1)
    gx = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "?"), gx // wg))
rows.sort()
adam = [i for i, r in enumerate(rows) if r[2].startswith("adam_kernel")]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 2
lo, hi = adam[-k - 1] + 1, adam[-k] + 1
step = rows[lo:hi]
t0 = rows[lo - 1][1]
print("step: %d kernels, previous Adam end -> this Adam end %.3f ms" % (len(step), (step[-1][1] - t0) / 1e6))
qs = sorted(set(r[3] for r in step))
last_end = {}
for s, e, n, q, b in step:
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = e
    print("%8.1f  q%-2s %s %7.1f us  gap %6.1f  blocks %-6d %s" % ((s - t0) / 1e3, q, " " * (2 * qs.index(q)), (e - s) / 1e3, gap, b, n))
