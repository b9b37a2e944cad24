#!/bin/bash
# PMC passes over one conv shape on the 32x32-tile kernel.  usage: t32_pmc.sh Ci Co H W B cfg
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R
OUT=$R/gpurun_out/t32_pmc; rm -rf $OUT; mkdir -p $OUT
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
P2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS"
P3="SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL"
P4="GRBM_GUI_ACTIVE GRBM_TA_BUSY TA_TA_BUSY TCP_PENDING_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do
  i=$((i+1))
  rocprofv3 --pmc $P --kernel-trace -d $OUT/p$i -o t --output-format csv -- python $R/tools/probes/t32_one.py "$@" > /dev/null 2>&1
done
python - <<PY
import csv, glob, collections
for i in range(1, 5):
    acc = collections.defaultdict(list)
    for f in glob.glob("$OUT/p%d/**/*counter_collection.csv" % i, recursive=True):
        for r in csv.DictReader(open(f)):
            if "t32" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        v = v[2:]
        print("%-32s %14.0f (per launch, mean of %d)" % (k, sum(v) / max(1, len(v)), len(v)))
PY
