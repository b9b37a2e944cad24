O=gpurun_out/r06n; mkdir -p $O
for i in 1; do
timeout 2400 python -m pytest tests -q -m gpu --no-header -p no:cacheprovider > $O/tests_$i.log 2>&1
echo "full $i rc=$?"; tail -2 $O/tests_$i.log | cut -c1-200
done
timeout 900 python -m pytest tests -q -m gpu -x --no-header -p no:cacheprovider -k "dp_gpu or graph or lanes or fullsize or model" > $O/mix.log 2>&1
echo "mix rc=$?"; tail -1 $O/mix.log | cut -c1-200
