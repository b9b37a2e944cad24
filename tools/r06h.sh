O=gpurun_out/r06h; mkdir -p $O
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
tail -3 $O/bench.err
bash tools/pmc_traffic.sh > $O/pmc.log 2>&1
python tools/pmc_traffic_summary.py gpurun_out/pmc_traffic $O/pmc_traffic.json > /dev/null 2>$O/pmc_sum.err
cat $O/pmc_sum.err | tail -3
python - <<PY
import json
d=json.load(open("$O/bench.json"))
print(d["ms_per_step"], d["ms_per_step_200"]); print(json.dumps(d["roofline"])[:900]); k=d["kernels"]
print({x:k[x] for x in ("sum_launch_ms","sum_alone_ms","launches_per_step","photo_fused_fwd","photo_fused_bwd")})
p=json.load(open("$O/pmc_traffic.json")); print({x:p[x] for x in ("photo_fused_fwd","photo_fused_bwd","conv3x3_halo") if x in p})
PY
