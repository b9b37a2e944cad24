O=gpurun_out/r06e; mkdir -p $O
for Q in 4 6 8 12; do
  GPU_MAX_HW_QUEUES=$Q python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-kernel-profile > $O/bench_q$Q.json 2> $O/bench_q$Q.err
  echo "Q=$Q $(python -c "import json;d=json.load(open('$O/bench_q$Q.json'));print(d['ms_per_step'])")"
done
GPU_MAX_HW_QUEUES=8 python tools/probes/step_marks.py > $O/marks_q8.txt 2>&1
GPU_MAX_HW_QUEUES=8 python tools/probes/dp_world1.py graph > $O/dp_auto_q8.txt 2>&1
grep -h "ms/step\|encoder pass" $O/dp_auto_q8.txt
tail -24 $O/marks_q8.txt
