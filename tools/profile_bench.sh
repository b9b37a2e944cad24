#!/bin/bash
# rocprofv3 kernel-trace summary of the benchmark command + PMC traffic passes (run on the GPU box from the repo root):
#   bash tools/profile_bench.sh <tag>   ->  gpurun_out/<tag>/{kernel_stats.md, pmc_traffic.json, bench.json}
#   BENCH_ARGS="--depth 50 --height 320 --width 1024 --batch 8" SKIP_PMC=1 bash tools/profile_bench.sh <tag>   (another config)
TAG=${1:-prof}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R
rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench --output-format csv -- python $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline $BENCH_ARGS > $OUT/bench.json 2> $OUT/bench.err
python - <<PY
import csv, glob, subprocess
rows = []
for f in glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open("$OUT/kernel_stats.md", "w") as o:
    o.write("rocprofv3 --kernel-trace --stats -- python bench.py --steps 50 --warmup 10 --no-cpu-baseline $BENCH_ARGS\n")
    o.write("(60 replayed steps + 3 eager profiling steps + warm-up/capture; durations in the profiled run)\n\n")
    o.write("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|\n")
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
        name = r["Name"]
        if name.startswith("_Z"):      # (rocprofv3 leaves some template instantiations mangled)
            try:
                name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip() or name
            except OSError:
                pass
        name = name.replace("(anonymous namespace)::", "").replace("void ", "")[:100]
        o.write("| %s | %s | %.3f | %.2f | %.1f |\n" % (name, r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                      float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
    o.write("\ntotal kernel time %.3f ms\n" % (tot / 1e6))
print(open("$OUT/kernel_stats.md").read()[:3000])
PY
[ -n "$SKIP_PMC" ] && exit 0
cd $R && bash tools/pmc_traffic.sh > /dev/null 2>&1
python tools/pmc_traffic_summary.py gpurun_out/pmc_traffic $OUT/pmc_traffic.json > /dev/null
cat $OUT/pmc_traffic.json | head -80
