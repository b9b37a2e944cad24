#!/bin/bash
# same-box A/B of environment switches: tools/gpu_ab.sh "A=1 B=0" "A=0 B=0" ...   (each: bench.py --steps 100, ms/step)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
: > gpurun_out/ab.txt
for rep in 1 2; do
for cfg in "$@"; do
  ms=$(env $cfg timeout 600 python bench.py --steps ${AB_STEPS:-100} --warmup 15 --no-cpu-baseline --no-kernel-profile $BENCH_ARGS 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.readline())['ms_per_step'])")
  echo "$cfg  ->  $ms ms" | tee -a gpurun_out/ab.txt
done
done
