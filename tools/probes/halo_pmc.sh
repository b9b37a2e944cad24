#!/bin/bash
# SQ / LDS / TA counters of the conv kernels on the four ResNet stage shapes (separate passes, kernel-trace only).
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R
CMD="python $R/tools/probes/conv_pmc.py"
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_WR TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/halo_pmc/p$i -o pmc --output-format csv -- $CMD > $R/gpurun_out/halo_pmc_p$i.log 2>&1
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$R/gpurun_out/halo_pmc/p*/pmc_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "conv3x3_halo" not in k and "wgrad3x3_halo" not in k:
            continue
        key = ("halo" if "conv3x3" in k else "wgrad") + " grid=" + r["Grid_Size"]
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key in sorted(acc):
    print(key)
    for c, v in sorted(acc[key].items()):
        print("   %-28s %14.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
