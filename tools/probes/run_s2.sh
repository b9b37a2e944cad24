cd /root/repo
timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_conv_fold_gpu.py -q -m gpu -x --no-header -p no:cacheprovider -k "stride2 or conv_fwd_bwd" 2>&1 | tail -15
bash tools/gpu_ab.sh "FSNET_AMD_HALO_S2=0" "FSNET_AMD_HALO_S2=1" "FSNET_AMD_HALO_S2=1 FSNET_AMD_S2_CO=16" "FSNET_AMD_HALO_S2=1 FSNET_AMD_S2_CO=32"
