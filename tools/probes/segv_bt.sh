#!/bin/bash
# native backtrace of an intermittent segfault in hipGraphLaunch during the GPU suite (run on the GPU box from the repo root)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out/segv; mkdir -p $O
for i in 1 2; do
  timeout 1500 /opt/rocm/bin/rocgdb -batch -ex "set pagination off" -ex "handle SIGSEGV stop print" -ex run -ex "bt 6" -ex "info registers" -ex "x/24i \$pc-60" -ex "x/6gx \$rdx" -ex "frame 1" -ex "info registers rbx r12 r13 r14 r15 rbp" \
      --args python -m pytest tests -q -m gpu -x --no-header -p no:cacheprovider -p no:faulthandler -k "dp_gpu or graph or lanes or fullsize or model" > $O/gdb_$i.log 2>&1
  if grep -q "SIGSEGV" $O/gdb_$i.log; then
    echo "== segfault in run $i"
    awk '/received signal SIGSEGV/{f=1} f' $O/gdb_$i.log | grep -v "^#[1-9][0-9]\|Thread 0x" | head -120
    break
  else
    echo "run $i: no segfault"; tail -2 $O/gdb_$i.log | cut -c1-200
  fi
done
