cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/late
python tools/probes/fisheye_late.py 2>&1 | tail -5
python -m pytest tests/test_fisheye_gpu.py tests/test_graph_gpu.py -x -q 2>&1 | tail -3
run() { n="$1"; shift; env $ENVV python tools/probes/ab_rt.py $SETS -- "$@" --no-cpu-baseline --no-kernel-profile 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', '$ENVV', '$SETS', d['ms_per_step'])"; }
for rep in 1 2; do
ENVV="A=1" SETS="" run base --steps 150 --warmup 20
ENVV="A=1" SETS="wgrad_balance=0" run base --steps 150 --warmup 20
ENVV="A=1" SETS="wgrad_balance=0 wgrad_flush_even=8" run base --steps 150 --warmup 20
ENVV="FSNET_AMD_LANES=1" SETS="" run lanes --steps 150 --warmup 20
ENVV="A=1" SETS="" run r50 --depth 50 --height 320 --width 1024 --batch 8 --steps 40 --warmup 10
ENVV="A=1" SETS="wgrad_flush=4" run r50 --depth 50 --height 320 --width 1024 --batch 8 --steps 40 --warmup 10
ENVV="A=1" SETS="wgrad_late=false" run r50 --depth 50 --height 320 --width 1024 --batch 8 --steps 40 --warmup 10
ENVV="A=1" SETS="" run fisheye --workload fisheye --steps 80 --warmup 10
ENVV="A=1" SETS="" run fp32 --dtype fp32 --steps 80 --warmup 10
ENVV="A=1" SETS="wgrad_balance=0" run fp32 --dtype fp32 --steps 80 --warmup 10
done | tee gpurun_out/late/sweep7.txt
