cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -s 2>&1 | grep -v "dist-packages" | tail -30 > gpurun_out/r02k_tests.log
cat gpurun_out/r02k_tests.log | cut -c1-330
