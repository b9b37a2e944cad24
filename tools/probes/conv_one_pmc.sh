#!/bin/bash
# kernel durations + SQ counters of one conv shape, resident-weights kernel on and off.  usage: conv_one_pmc.sh Ci Co H W B
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R
for res in 1 0; do
  export FSNET_AMD_HALO_RES=$res
  out=$R/gpurun_out/conv_one/res$res
  rocprofv3 --kernel-trace --stats -d $out/trace -o t --output-format csv -- python $R/tools/probes/conv_one.py "$@" > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $out/pmc1 -o p --output-format csv -- python $R/tools/probes/conv_one.py "$@" > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY -d $out/pmc2 -o p --output-format csv -- python $R/tools/probes/conv_one.py "$@" > /dev/null 2>&1
  echo "== FSNET_AMD_HALO_RES=$res"
  python - <<PY
import csv, glob, collections
for f in glob.glob("$out/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv3x3" in r["Name"]:
            print("  %-60s calls %s avg %.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$out/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv3x3" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:50] + " grid=" + r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(" ", k)
    for c, v in sorted(acc[k].items()):
        print("     %-24s %12.0f" % (c, sum(v) / len(v)))
PY
done
