O=gpurun_out/r06d; mkdir -p $O
timeout 1500 python -m pytest tests/test_dp_gpu.py tests/test_dp2_gpu.py tests/test_dp_capture_failure_gpu.py -q -m gpu -x --no-header -p no:cacheprovider > $O/tests.log 2>&1
echo "pytest rc=$?" >> $O/tests.log
tail -12 $O/tests.log
python tools/probes/dp_world1.py nodp graph > $O/dp_nodp.txt 2>&1
python tools/probes/dp_world1.py graph > $O/dp_auto.txt 2>&1
grep -h "ms/step\|encoder pass" $O/dp_*.txt
