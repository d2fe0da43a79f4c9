# GPU call r06s: encoder GPU tests after the plan-query fixes + the new chain on/off parity test
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_encoders.py tests/test_gpu_parsing.py tests/test_gpu_schedule.py -x -q -m gpu 2>&1 | tail -8
