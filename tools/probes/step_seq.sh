#!/bin/bash
# kernel trace of a short benchmark run (replays only) + the sequence of one replayed step
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/step_seq
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R
rocprofv3 --kernel-trace -d $OUT -o t --output-format csv -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-profile "$@" > $OUT/bench.json 2> $OUT/bench.err
tail -1 $OUT/bench.json | cut -c1-300
F=$(ls $OUT/*kernel_trace.csv $OUT/*/*kernel_trace.csv 2>/dev/null | head -1)
python $R/tools/step_sequence.py $F 3 > $OUT/sequence.txt
python $R/tools/step_timeline.py $F 3 > $OUT/timeline.txt
rm -f $OUT/*.csv $OUT/*/*.csv
head -30 $OUT/timeline.txt
