#!/bin/bash
# kernel durations + SQ / L2 counters of one 1x1 conv shape.  usage: conv1x1_pmc.sh Ci Co H W B [stride]
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R
out=$R/gpurun_out/conv1x1_pmc
rm -rf $out
P="python $R/tools/probes/conv1x1_one.py $@"
rocprofv3 --kernel-trace --stats -d $out/trace -o t --output-format csv -- $P > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $out/pmc1 -o p --output-format csv -- $P > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY -d $out/pmc2 -o p --output-format csv -- $P > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/pmc3 -o p --output-format csv -- $P > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/pmc4 -o p --output-format csv -- $P > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d $out/pmc5 -o p --output-format csv -- $P > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc TA_BUSY_avr TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum -d $out/pmc6 -o p --output-format csv -- $P > /dev/null 2>&1
python - <<PY
import csv, glob, collections
for f in glob.glob("$out/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv" in r["Name"]:
            print("  %-70s calls %s avg %.1f us min %.1f" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$out/pmc*/**/*counter_collection.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    seen = collections.Counter()
    for r in rows:
        if "conv" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:60] + " grid=" + r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(" ", k)
    for c, v in sorted(acc[k].items()):
        # the launches come in three groups of six: no statistics, statistics, dgrad
        n = len(v) // 6 if len(v) >= 6 else 1
        print("     %-28s" % c, " ".join("%12.0f" % (sum(v[i * 6:(i + 1) * 6]) / max(1, len(v[i * 6:(i + 1) * 6]))) for i in range((len(v) + 5) // 6)))
PY
