cd $GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu -x 2>&1 | grep -v Warning | tail -6
