cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/late
run() { n="$1"; shift; env $ENVV python tools/probes/ab_rt.py $SETS -- "$@" --no-cpu-baseline --no-kernel-profile 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', '$SETS', d['ms_per_step'])"; }
for rep in 1 2; do
for S in "" "wgrad_share_decoder=true" "wgrad_share_decoder=true wgrad_share_stage=-1" "wgrad_share_decoder=true wgrad_share_stage=3" "wgrad_share_stage=-1"; do
ENVV="A=1" SETS="$S" run base --steps 150 --warmup 20
done
done | tee gpurun_out/late/sweep13.txt
