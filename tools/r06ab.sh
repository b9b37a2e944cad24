cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/late
run() { n="$1"; shift; env $ENVV python tools/probes/ab_rt.py $SETS -- "$@" --no-cpu-baseline --no-kernel-profile 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', '$SETS', d['ms_per_step'])"; }
for rep in 1 2 3; do
for S in "chain_end_nop=false" ""; do
ENVV="A=1" SETS="$S" run base --steps 150 --warmup 20
done
done | tee gpurun_out/late/sweep11.txt
for S in "chain_end_nop=false" ""; do
ENVV="A=1" SETS="$S" run r50 --depth 50 --height 320 --width 1024 --batch 8 --steps 40 --warmup 10
ENVV="A=1" SETS="$S" run fisheye --workload fisheye --steps 80 --warmup 10
done | tee -a gpurun_out/late/sweep11.txt
mkdir -p /tmp/dotf && cd /tmp/dotf && DEBUG_HIP_GRAPH_DOT_PRINT=1 python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 5 --no-cpu-baseline --no-kernel-profile > out.txt 2> err.txt; python $GRAFT_REPO_ROOT/tools/probes/graph_streams.py $(ls -S graph_*dot_print* | head -1) | tee $GRAFT_REPO_ROOT/gpurun_out/late/streams_final.txt
