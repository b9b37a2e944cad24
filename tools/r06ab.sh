cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/late
python -m pytest tests/test_graph_gpu.py tests/test_fisheye_gpu.py tests/test_dp_standin_capture_gpu.py tests/test_dp_capture_failure_gpu.py -x -q 2>&1 | grep -v Warning | tail -4
run() { n="$1"; shift; env $ENVV python tools/probes/ab_rt.py $SETS -- "$@" --no-cpu-baseline --no-kernel-profile 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', '$SETS', d['ms_per_step'])"; }
for rep in 1 2; do
ENVV="A=1" SETS="" run base --steps 150 --warmup 20
done | tee gpurun_out/late/sweep12.txt
ENVV="A=1" SETS="" run r50 --depth 50 --height 320 --width 1024 --batch 8 --steps 40 --warmup 10 | tee -a gpurun_out/late/sweep12.txt
ENVV="A=1" SETS="" run fisheye --workload fisheye --steps 80 --warmup 10 | tee -a gpurun_out/late/sweep12.txt
for rep in 1 2; do
python tools/probes/dp_world1.py graph nodp 2>&1 | grep "ms/step"
for L in 0 1; do for Wg in inline companion tail; do
echo "lanes=$L wgrad=$Wg fakecomm: $(FSNET_AMD_LANES=$L FSNET_AMD_DP_WGRAD=$Wg python tools/probes/dp_world1.py graph fakecomm 2>&1 | grep 'ms/step')"
done; done
done | tee gpurun_out/late/dp4.txt
mkdir -p /tmp/dotf && cd /tmp/dotf && DEBUG_HIP_GRAPH_DOT_PRINT=1 python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 5 --no-cpu-baseline --no-kernel-profile > out.txt 2> err.txt; python $GRAFT_REPO_ROOT/tools/probes/graph_streams.py $(ls -S graph_*dot_print* | head -1) | tee $GRAFT_REPO_ROOT/gpurun_out/late/streams_final2.txt
mkdir -p /tmp/dot_c && cd /tmp/dot_c && FSNET_AMD_LANES=0 FSNET_AMD_DP_WGRAD=companion DEBUG_HIP_GRAPH_DOT_PRINT=1 python $GRAFT_REPO_ROOT/tools/probes/dp_world1.py graph fakecomm > out.txt 2> err.txt; python $GRAFT_REPO_ROOT/tools/probes/graph_streams.py $(ls -S graph_*dot_print* | head -1) | tee $GRAFT_REPO_ROOT/gpurun_out/late/streams_dpfake_companion2.txt
