# Round check on the GPU box (run through `gpurun -- bash gpurun_cmd.sh` from the repo root):
# GPU parity suite, smoke, the default bench line.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --durations=15 > gpurun_out/tests_gpu.log 2>&1; tail -40 gpurun_out/tests_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
