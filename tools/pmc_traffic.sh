#!/bin/bash
# HBM traffic of the conv kernels from PMC counters: two separate rocprofv3 passes (FETCH_SIZE needs 3 TCC
# slots, WRITE_SIZE 2), kernel-trace only; a third pass counts wave-level vector instructions (SQ_INSTS_VALU).  Run on the GPU box from the repo root:
#   bash tools/pmc_traffic.sh   ->   gpurun_out/pmc_traffic/{fetch,write}/pmc_counter_collection.csv
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R
export FSNET_AMD_GRAPH=0   # per-dispatch counters: eager launches (the same kernels the graph replays)
CMD="python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-profile $BENCH_ARGS"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_traffic/fetch -o pmc --output-format csv -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc_traffic/write -o pmc --output-format csv -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES -d $R/gpurun_out/pmc_traffic/valu -o pmc --output-format csv -- $CMD > /dev/null 2>&1
ls $R/gpurun_out/pmc_traffic/*
