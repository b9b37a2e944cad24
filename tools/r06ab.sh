cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/late
python -m pytest tests/test_graph_gpu.py tests/test_fisheye_gpu.py -x -q 2>&1 | grep -v Warning | tail -3
run() { n="$1"; shift; env $ENVV python tools/probes/ab_rt.py $SETS -- "$@" --no-cpu-baseline --no-kernel-profile 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', '$SETS', d['ms_per_step'])"; }
for rep in 1 2 3; do
for S in "wgrad_balance=0" ""; do
ENVV="A=1" SETS="$S" run base --steps 150 --warmup 20
ENVV="A=1" SETS="$S" run r50 --depth 50 --height 320 --width 1024 --batch 8 --steps 40 --warmup 10
ENVV="A=1" SETS="$S" run fp32 --dtype fp32 --steps 60 --warmup 10
done
done | tee gpurun_out/late/sweep10.txt
