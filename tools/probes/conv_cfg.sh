#!/bin/bash
# kernel-trace durations of one conv shape under each experimental halo configuration (FSNET_AMD_HALO_CFG)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R
CFGS=${CFGS:-"0 1 2 3 4 5 6"}
for c in $CFGS; do
  export FSNET_AMD_HALO_CFG=$c
  out=$R/gpurun_out/conv_cfg/c$c
  rm -rf $out
  rocprofv3 --kernel-trace -d $out -o t --output-format csv -- python $R/tools/probes/conv_one.py "$@" > /dev/null 2>&1
  python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$out/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv3x3" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][33:62]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in acc.items():
    print("cfg=$c %-30s fwd %s | dgrad %s" % (k, " ".join("%.1f" % x for x in v[2:7]), " ".join("%.1f" % x for x in v[-5:])))
PY
done
