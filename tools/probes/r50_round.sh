#!/bin/bash
# ResNet-50 @320x1024 measurement set (run on the GPU box from the repo root):  bash tools/probes/r50_round.sh <tag>
#   gpurun_out/<tag>/{r50_layers.md, conv1x1_shapes.txt, conv1x1_pmc.txt, r50_pmc_traffic.txt, bench_r50.json}
TAG=${1:-r50}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
A="--depth 50 --height 320 --width 1024 --batch 8"
python bench.py $A --steps 30 --warmup 8 --no-cpu-baseline > $OUT/bench_r50.json 2> $OUT/bench_r50.err
python tools/layer_table.py $A > $OUT/r50_layers.md 2> /dev/null
for g in 1 0; do
  echo "== FSNET_AMD_1X1_GEMM=$g FSNET_AMD_WGRAD_1X1=$g  (batch 16: the stacked pose pairs)"
  FSNET_AMD_1X1_GEMM=$g FSNET_AMD_WGRAD_1X1=$g python tools/probes/conv1x1_shapes.py 16 2>&1 | tail -12
done > $OUT/conv1x1_shapes.txt
bash tools/probes/conv1x1_pmc.sh 256 1024 20 64 16 > $OUT/conv1x1_pmc.txt 2>&1
BENCH_ARGS="$A" bash tools/pmc_traffic.sh > /dev/null 2>&1
python tools/pmc_traffic_by_kernel.py gpurun_out/pmc_traffic > $OUT/r50_pmc_traffic.txt
rm -rf gpurun_out/pmc_traffic gpurun_out/conv1x1_pmc
ls -la $OUT
