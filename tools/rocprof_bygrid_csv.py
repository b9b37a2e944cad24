"""rocprofv3 --kernel-trace CSV grouped by (kernel, grid): calls, avg / min duration, share of the kernel time.
python tools/rocprof_bygrid_csv.py <..._kernel_trace.csv> [pattern]"""
import csv, re, sys
pat = sys.argv[2] if len(sys.argv) > 2 else ""
agg, tot = {}, 0
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if pat not in n:
        continue
    m = re.search(r"(\w+_kernel)", n)
    tpl = re.search(r"_kernelI(\w+?)EEv", n)
    k = ((m.group(1) if m else n[:40]) + ("<%s>" % tpl.group(1) if tpl else ""),
         (int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)) * (int(r["Grid_Size_Y"]) // max(int(r["Workgroup_Size_Y"]), 1)) * (int(r["Grid_Size_Z"]) // max(int(r["Workgroup_Size_Z"]), 1)), int(r["Workgroup_Size_X"]))
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    a = agg.setdefault(k, [0, 0, 1 << 60])
    a[0] += 1; a[1] += d; a[2] = min(a[2], d); tot += d
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-64s blocks %6d x %4d  n=%5d  avg %8.2f us  min %7.2f  %5.2f %%" % (k[0][:64], k[1], k[2], a[0], a[1] / a[0] / 1e3, a[2] / 1e3, 100.0 * a[1] / tot))
