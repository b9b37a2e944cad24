#!/bin/bash
# end-of-round measurement set on the GPU box (from the repo root):  bash tools/final_profiles.sh <tag>
#   gpurun_out/<tag>/{kernel_stats.md, bench.json (profiled), pmc_traffic.json, bench_n1.json (unprofiled, 200 steps),
#                      step_marks.txt, conv_pmc.txt, wgrad_pmc.txt, bench_r50.json, bench_fisheye.json}
TAG=${1:-final}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
bash tools/profile_bench.sh $TAG > /dev/null 2>&1
python tools/probes/step_marks.py > $OUT/step_marks.txt 2>&1
bash tools/probes/conv_pmc2.sh 64 64 48 160 12 > $OUT/conv_pmc.txt 2>&1
bash tools/probes/wg_pmc.sh 64 64 48 160 12 > $OUT/wgrad_pmc.txt 2>&1
python bench.py --depth 50 --height 320 --width 1024 --batch 8 --steps 30 --warmup 8 --no-cpu-baseline > $OUT/bench_r50.json 2> $OUT/bench_r50.err
python bench.py --workload fisheye --steps 60 --warmup 10 --no-cpu-baseline > $OUT/bench_fisheye.json 2> $OUT/bench_fisheye.err
python tools/rocprof_bygrid_csv.py $OUT/trace/bench_kernel_trace.csv > $OUT/kernels_by_grid.txt 2>/dev/null
ls -la $OUT
