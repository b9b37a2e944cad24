cd $GRAFT_REPO_ROOT
python -m pytest tests/test_dp_standin_capture_gpu.py -x -q 2>&1 | grep -v Warning | tail -30
