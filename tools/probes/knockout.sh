#!/bin/bash
# step time with each kernel family dropped in turn (tools/probes/knockout.py)
for s in none fs_conv3x3_halo fs_conv_wgrad fs_photo_fused_fwd,fs_photo_fused_bwd,fs_photo_identity fs_bn_bwd_apply fs_conv_igemm fs_bn_apply \
         fs_bn_bwd_reduce fs_maxpool_fwd,fs_maxpool_bwd fs_conv_stem fs_adam_step,fs_sumsq fs_depth_head_fwd_multi,fs_depth_head_bwd_multi \
         fs_upcat_pad_fwd,fs_upcat_pad_bwd fs_conv1x1 fs_channel_sum fs_smooth_fwd,fs_smooth_bwd,fs_color_pyramid none; do
  printf "%-60s " "$s"; timeout 120 python tools/probes/knockout.py $s --steps 60 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json
try:
    d=json.loads(sys.stdin.read()); print(d['ms_per_step'])
except Exception as e: print('fail', e)"
done
