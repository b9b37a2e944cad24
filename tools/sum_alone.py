"""Per kernel family: launches per step, sum of in-situ average and of minimum ("alone") durations per step, from a
tools/rocprof_bygrid_csv.py table.   python tools/sum_alone.py <kernels_by_grid.txt>
(steps = number of adam_kernel launches in the trace; eager profiling / warm-up steps of the same run are averaged in)"""
import collections
import re
import sys

fam = collections.OrderedDict()
rows = []
for l in open(sys.argv[1]):
    m = re.match(r'(\S.*?)\s+blocks\s+(\d+) x\s+(\d+)\s+n=\s*(\d+)\s+avg\s+([\d.]+) us\s+min\s+([\d.]+)', l)
    if not m:
        continue
    name = re.sub(r'_ZN12_GLOBAL__N_1\d+', '', m.group(1))
    name = re.sub(r'[<(].*', '', name)
    name = re.sub(r'I(DF16b|f)L.*|I(DF16b|f)E.*', '', name)
    rows.append((name, int(m.group(4)), float(m.group(5)), float(m.group(6))))
steps = float(sum(n for name, n, _, _ in rows if name.startswith("adam_kernel")) or 1)
for name, n, avg, mn in rows:
    f = fam.setdefault(name, [0.0, 0.0, 0.0])
    f[0] += n / steps; f[1] += n * avg / steps; f[2] += n * mn / steps
print("steps in the trace: %d" % steps)
for k, v in sorted(fam.items(), key=lambda kv: -kv[1][2]):
    print("%-34s launches/step %6.1f   sum of averages %8.1f us   sum of minima (alone) %8.1f us" % (k, v[0], v[1], v[2]))
print("TOTAL  launches/step %.1f   sum of averages %.3f ms   sum of minima (alone) %.3f ms" % (
    sum(v[0] for v in fam.values()), sum(v[1] for v in fam.values()) / 1e3, sum(v[2] for v in fam.values()) / 1e3))
