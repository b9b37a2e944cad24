#!/bin/bash
# kernel-trace durations (rocprofv3) of the conv kernels launched by a python command.  usage: ktrace.sh <tag> <python args...>
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R
OUT=$R/gpurun_out/ktrace/$TAG; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace -d $OUT -o t --output-format csv -- python $R/"$@" > $OUT/stdout.txt 2>&1
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "conv3x3" in n or "wgrad" in n or "copy" in n.lower():
            acc[n[:110]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in acc.items():
    w = sorted(v[2:]) if len(v) > 4 else sorted(v)
    print("$TAG %-112s n=%3d median %.1f us  min %.1f" % (k, len(v), w[len(w) // 2], w[0]))
PY
