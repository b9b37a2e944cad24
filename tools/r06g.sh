O=gpurun_out/r06g; mkdir -p $O
python tools/probes/photo_skip_rate.py > $O/skip_rate.txt 2>&1
grep -h "identity wins" $O/skip_rate.txt
timeout 900 python -m pytest tests/test_trains_gpu.py -q -m gpu -x --no-header -p no:cacheprovider -s > $O/trains.log 2>&1
grep -h "ResNet-\|passed\|failed\|Error" $O/trains.log | tail -12
