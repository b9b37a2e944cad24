"""HBM traffic per kernel family and step from tools/pmc_traffic.sh output (FETCH_SIZE / WRITE_SIZE in KB, eager steps).
    python tools/pmc_traffic_by_kernel.py [gpurun_out/pmc_traffic] [steps]
FETCH_SIZE is shown raw and doubled (gfx950 reports half of wide coalesced reads: MI355X_MICROARCH.md, HBM section)."""
import collections
import csv
import re
import sys

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc_traffic"
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 0


def load(path, counter):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"])
        name = re.sub(r"[<(].*", "", name)
        name = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", name)            # (mangled template instantiations)
        name = re.sub(r"_kernelI.*|_kernel$", "_kernel", name)
        acc[name][0] += 1
        acc[name][1] += float(r["Counter_Value"])
    return acc


f = load(root + "/fetch/pmc_counter_collection.csv", "FETCH_SIZE")
w = load(root + "/write/pmc_counter_collection.csv", "WRITE_SIZE")
if not steps:
    steps = float(f.get("adam_kernel", [1])[0] or 1)
rows = []
for k in set(f) | set(w):
    fk, wk = f.get(k, [0, 0.0]), w.get(k, [0, 0.0])
    rows.append((2 * fk[1] + wk[1], k, max(fk[0], wk[0]) / steps, fk[1] / steps / 1024, wk[1] / steps / 1024))
rows.sort(reverse=True)
print("eager steps sampled: %d   (MB per step; fetch raw | x2, write)" % steps)
tot = [0.0, 0.0]
for _, k, n, fm, wm in rows:
    tot[0] += fm; tot[1] += wm
    if 2 * fm + wm >= 5:
        print("%-34s launches/step %6.1f   fetch %8.1f | %8.1f MB   write %8.1f MB" % (k[:34], n, fm, 2 * fm, wm))
print("TOTAL  fetch %.1f | %.1f MB   write %.1f MB   -> %.1f GB per step (fetch x2 + write)" % (tot[0], 2 * tot[0], tot[1], (2 * tot[0] + tot[1]) / 1024))
