#!/bin/bash
# native backtrace of the intermittent abort in the captured data-parallel step test
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for i in 1 2 3 4 5 6; do
  timeout 300 /opt/rocm/bin/rocgdb -batch -ex "set pagination off" -ex "handle SIGABRT stop print" -ex run -ex "thread apply all bt 25" \
      --args python -m pytest tests/test_dp_gpu.py -x -q -k hipgraph > gpurun_out/dp_gdb_$i.log 2>&1
  if grep -q "SIGABRT" gpurun_out/dp_gdb_$i.log; then
    echo "== abort in run $i"
    grep -n "SIGABRT" gpurun_out/dp_gdb_$i.log | head -3
    awk '/received signal SIGABRT/{f=1} f' gpurun_out/dp_gdb_$i.log | grep -A28 "^Thread 1 \|(Thread .*LWP" | head -120
    break
  fi
done
