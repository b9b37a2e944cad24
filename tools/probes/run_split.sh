cd /root/repo
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_graph_gpu.py tests/test_ops_gpu.py tests/test_fisheye_gpu.py tests/test_distill_gpu.py tests/test_dp_gpu.py tests/test_dp2_gpu.py tests/test_fullsize_gpu.py -q -m gpu -x --no-header -p no:cacheprovider 2>&1 | tail -15
bash tools/gpu_ab.sh "FSNET_AMD_SPLIT_PHOTO_BWD=0" "FSNET_AMD_SPLIT_PHOTO_BWD=1"
