O=gpurun_out/r06f; mkdir -p $O
python tools/probes/scene_train2.py depth=18 pre=400 steps=800 seed=21 > $O/r18_two_s.txt 2>&1
python tools/probes/scene_train2.py depth=18 pre=400 steps=800 seed=5 > $O/r18_two_s5.txt 2>&1
python tools/probes/scene_train2.py depth=18 pre=400 steps=800 seed=7 > $O/r18_two_s7.txt 2>&1
python tools/probes/scene_train2.py depth=50 pre=600 steps=800 seed=21 > $O/r50_two_s.txt 2>&1
python tools/probes/scene_train2.py depth=50 pre=600 steps=800 seed=5 > $O/r50_two_s5.txt 2>&1
grep -h "^R" $O/r18_two_s.txt $O/r18_two_s5.txt $O/r18_two_s7.txt $O/r50_two_s.txt $O/r50_two_s5.txt
