set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/late
for rep in 1 2; do
for v in 0 1; do
  FSNET_AMD_WGRAD_LATE=$v python bench.py --steps 150 --warmup 20 --no-cpu-baseline --no-kernel-profile 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('late=$v', d['ms_per_step'], d['value'])"
done; done | tee gpurun_out/late/ab.txt
mkdir -p /tmp/dot1 && cd /tmp/dot1 && DEBUG_HIP_GRAPH_DOT_PRINT=1 python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 5 --no-cpu-baseline --no-kernel-profile > out.txt 2> err.txt; ls; cp graph_*dot_print* $GRAFT_REPO_ROOT/gpurun_out/late/ ; cd $GRAFT_REPO_ROOT; python tools/probes/graph_streams.py gpurun_out/late/graph_*dot_print* | tee gpurun_out/late/streams.txt
