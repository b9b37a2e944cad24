"""Timeline of ONE replayed training step from a rocprofv3 --kernel-trace CSV: wall span, busy union, average
concurrency, idle gaps, phase boundaries (photometric forward / backward, Adam) and the kernels on the GPU during the
longest single-kernel stretches.    python tools/step_timeline.py <kernel_trace.csv> [step_index_from_end]"""
import collections
import csv
import sys


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    if n.startswith("_ZN12_GLOBAL__N_1"):
        n = n[17:].lstrip("0123456789")
    return n[:44]


def main():
    rows = []
    for r in csv.DictReader(open(sys.argv[1])):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]),
                     r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
    rows.sort()
    adam = [i for i, r in enumerate(rows) if r[2].startswith("adam_kernel")]
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    lo, hi = adam[-k - 1] + 1, adam[-k] + 1
    step = rows[lo:hi]
    t0, t1 = step[0][0], max(r[1] for r in step)
    print("step: %d kernels, wall %.3f ms (previous Adam end -> this Adam end %.3f ms)"
          % (len(step), (t1 - t0) / 1e6, (step[-1][1] - rows[lo - 1][1]) / 1e6))
    ev = []
    for s, e, *_ in step:
        ev.append((s, 1)); ev.append((e, -1))
    ev.sort()
    busy = 0; conc_time = collections.Counter(); n = 0; last = ev[0][0]; gaps = []
    for t, d in ev:
        if t > last:
            conc_time[n] += t - last
            if n == 0:
                gaps.append((t - last, last))
        n += d; last = t
    tot = sum(conc_time.values())
    print("kernel time summed %.3f ms; GPU idle %.3f ms (%d gaps); time by number of kernels in flight:"
          % (sum(e - s for s, e, *_ in step) / 1e6, conc_time[0] / 1e6, len(gaps)))
    for c in sorted(conc_time):
        print("   %d: %.3f ms (%.1f %%)" % (c, conc_time[c] / 1e6, 100 * conc_time[c] / tot))
    marks = [r for r in step if r[2].startswith(("photo_fused_fwd", "photo_fused_bwd", "adam_kernel", "sumsq", "maxpool_fwd",
                                                 "conv_stem", "wgrad_stem", "depth_head_bwd", "depth_head_fwd"))]
    print("phase markers (ms from step start):")
    for s, e, nme, q, st in marks:
        print("   %7.3f .. %7.3f  %s" % ((s - t0) / 1e6, (e - t0) / 1e6, nme))
    # time with exactly one kernel in flight, by kernel
    solo = collections.Counter()
    active = {}
    ev2 = []
    for i, (s, e, nme, *_rest) in enumerate(step):
        ev2.append((s, 1, i)); ev2.append((e, -1, i))
    ev2.sort()
    last = ev2[0][0]
    for t, d, i in ev2:
        if len(active) == 1 and t > last:
            solo[step[next(iter(active))][2]] += t - last
        if d == 1:
            active[i] = 1
        else:
            active.pop(i, None)
        last = t
    print("time alone on the GPU, by kernel (top 15):")
    for nme, v in solo.most_common(15):
        print("   %7.3f ms  %s" % (v / 1e6, nme))
    print("largest idle gaps:")
    for g, at in sorted(gaps, reverse=True)[:8]:
        print("   %6.1f us at %.3f ms" % (g / 1e3, (at - t0) / 1e6))
    byq = collections.defaultdict(float)
    for s, e, nme, q, st in step:
        byq[(q, st)] += (e - s) / 1e6
    print("busy time per (queue, stream):", {k: round(v, 3) for k, v in byq.items()})


main()
