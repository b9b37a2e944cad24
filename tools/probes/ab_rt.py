"""bench.py with RT attributes overridden: A/B of the engine's placement constants on one box.
    python tools/probes/ab_rt.py wgrad_flush_even=8 wgrad_balance=0 -- --steps 150 --warmup 20 --no-cpu-baseline --no-kernel-profile"""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
cut = sys.argv.index("--") if "--" in sys.argv else len(sys.argv)
sets, rest = sys.argv[1:cut], sys.argv[cut + 1:]
from fsnet_amd.engine.runtime import RT
for kv in sets:
    k, v = kv.split("=")
    setattr(RT, k, json.loads(v))
sys.argv = ["bench.py"] + rest
import bench
bench.main()
