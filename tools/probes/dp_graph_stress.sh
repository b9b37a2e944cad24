# usage: dp_graph_stress.sh N [ENV=VAL ...] : run the world-1 graph-captured data-parallel step N times, count aborts
N=$1; shift
export PYTHONPATH=$PWD FSNET_AMD_GRAPH_DP=1
ok=0; bad=0
for i in $(seq 1 $N); do
  if env "$@" timeout 200 python tools/probes/dp_world1.py graph > /tmp/o.txt 2>&1; then ok=$((ok+1)); else bad=$((bad+1)); grep -m1 "terminated with exception" /tmp/o.txt | cut -c1-220; fi
done
echo "$@ OK=$ok BAD=$bad"; grep "dp=" /tmp/o.txt
