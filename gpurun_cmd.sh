cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -q 2>&1 | tail -12
