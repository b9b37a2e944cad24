#!/bin/bash
# lanes on / off: step marks + per-grid kernel tables of both (GPU box, from the repo root)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/lanes_ab
mkdir -p $OUT
cd $R
for L in 1 0; do
  FSNET_AMD_LANES=$L python tools/probes/step_marks.py > $OUT/marks_lanes$L.txt 2>&1
  ( cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R && FSNET_AMD_LANES=$L rocprofv3 --kernel-trace --stats -d $OUT/trace$L -o bench --output-format csv -- python $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline > $OUT/bench$L.json 2> $OUT/bench$L.err )
  python tools/rocprof_bygrid_csv.py $(find $OUT/trace$L -name '*kernel_trace.csv' | head -1) > $OUT/bygrid_lanes$L.txt 2>/dev/null
  rm -rf $OUT/trace$L
done
tail -25 $OUT/marks_lanes1.txt
