#!/bin/bash
# kernel trace of a short benchmark run, per (kernel, grid) table
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/step_grid
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R
rocprofv3 --kernel-trace -d $OUT -o t --output-format csv -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline "$@" > $OUT/bench.json 2> $OUT/bench.err
tail -1 $OUT/bench.json | cut -c1-200
python $R/tools/probes/kgrid.py $(ls $OUT/*kernel_trace.csv $OUT/*/*kernel_trace.csv 2>/dev/null | head -1) 30 > $OUT/kgrid.txt
head -70 $OUT/kgrid.txt
