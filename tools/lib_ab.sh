#!/bin/bash
# same-box A/B of two builds of the library: tools/lib_ab.sh fsnet_amd/lib/var/a.so fsnet_amd/lib/var/b.so  (bench.py ms/step, two rounds)
cd "$(dirname "$0")/.."
cp fsnet_amd/lib/libfsnet_hip.so /tmp/lib_keep.so
for rep in 1 2; do
for v in "$@"; do
  cp $v fsnet_amd/lib/libfsnet_hip.so
  ms=$(timeout 600 python bench.py --steps ${AB_STEPS:-100} --warmup 15 --no-cpu-baseline --no-kernel-profile $BENCH_ARGS 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.readline())['ms_per_step'])")
  echo "$v  ->  $ms ms"
done
done
cp /tmp/lib_keep.so fsnet_amd/lib/libfsnet_hip.so
