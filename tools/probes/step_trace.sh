#!/bin/bash
# kernel trace of a short benchmark run + the one-step timeline summary
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/step_trace
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R
rocprofv3 --kernel-trace -d $OUT -o t --output-format csv -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > $OUT/bench.json 2> $OUT/bench.err
tail -1 $OUT/bench.json | cut -c1-300
python $R/tools/step_timeline.py $(ls $OUT/*kernel_trace.csv $OUT/*/*kernel_trace.csv 2>/dev/null | head -1) | tee $OUT/timeline.txt
