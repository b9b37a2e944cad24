#!/bin/bash
# kernel-trace durations of one conv shape (forward with / without the fused BatchNorm statistics, data gradient)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R
for ns in "" 1 pm0; do
  export NOSTATS=$ns; unset FSNET_AMD_HALO_PIXMAJOR
  if [ "$ns" = pm0 ]; then export NOSTATS= FSNET_AMD_HALO_PIXMAJOR=0; fi
  out=$R/gpurun_out/conv_trace/ns$ns
  rocprofv3 --kernel-trace -d $out -o t --output-format csv -- python $R/tools/probes/conv_one.py "$@" > /dev/null 2>&1
  python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$out/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][14:70]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in acc.items():
    print("NOSTATS=%s %-56s n=%d first6 %s" % ("$ns", k, len(v), " ".join("%.1f" % x for x in v[:7])), "| last6", " ".join("%.1f" % x for x in v[-6:]))
PY
done
