O=gpurun_out/r06c; mkdir -p $O
python tools/probes/step_marks.py > $O/marks_plain.txt 2>&1
FSNET_AMD_LANES=0 python tools/probes/step_marks.py dp > $O/marks_dp_chains.txt 2>&1
FSNET_AMD_LANES=1 python tools/probes/step_marks.py dp > $O/marks_dp_lanes.txt 2>&1
FSNET_AMD_LANES=1 python tools/probes/step_marks.py > $O/marks_lanes.txt 2>&1
tail -32 $O/marks_plain.txt; tail -32 $O/marks_dp_chains.txt
