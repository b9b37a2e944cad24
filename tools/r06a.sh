set -x
O=gpurun_out/r06a; mkdir -p $O
python bench.py --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_n1.json 2> $O/bench_n1.err
python tools/probes/dp_world1.py nodp graph > $O/dp_nodp.txt 2>&1
FSNET_AMD_LANES=0 python tools/probes/dp_world1.py graph > $O/dp_chains.txt 2>&1
FSNET_AMD_LANES=1 python tools/probes/dp_world1.py graph > $O/dp_lanes.txt 2>&1
tail -2 $O/dp_*.txt
cat $O/bench_n1.json | head -c 600
