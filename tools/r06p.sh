TAG=${1:-r06p}
O=gpurun_out/$TAG; mkdir -p $O
SKIP_PMC=1 bash tools/profile_bench.sh $TAG > /dev/null 2>&1
python tools/rocprof_bygrid_csv.py $(find $O/trace -name '*kernel_trace.csv' | head -1) > $O/kernels_by_grid.txt 2>/dev/null
python tools/sum_alone.py $O/kernels_by_grid.txt > $O/sum_alone_by_family.txt
rm -rf $O/trace
cat $O/sum_alone_by_family.txt
python -c "
import json;d=json.load(open('$O/bench.json'));print(d['ms_per_step'], d['kernels']['launches_per_step'])"
