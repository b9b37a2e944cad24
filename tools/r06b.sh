set -x
O=gpurun_out/r06b; mkdir -p $O
timeout 1200 python -m pytest tests/test_dp_gpu.py tests/test_dp2_gpu.py tests/test_dp_capture_failure_gpu.py tests/test_graph_gpu.py tests/test_lanes_gpu.py -q -m gpu -x --no-header -p no:cacheprovider > $O/tests.log 2>&1
echo "pytest rc=$?" >> $O/tests.log
tail -30 $O/tests.log
python tools/probes/dp_world1.py nodp graph > $O/dp_nodp.txt 2>&1
python tools/probes/dp_world1.py graph > $O/dp_auto.txt 2>&1
FSNET_AMD_DP_WGRAD=companion python tools/probes/dp_world1.py graph > $O/dp_auto_companion.txt 2>&1
FSNET_AMD_DP_PACK_OVERLAP=0 python tools/probes/dp_world1.py graph > $O/dp_auto_nopack.txt 2>&1
grep -h "ms/step\|encoder pass" $O/dp_*.txt
