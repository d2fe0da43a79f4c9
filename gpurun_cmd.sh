# GPU call r05a (run through `gpurun -- bash gpurun_cmd.sh` from the repo root): the 32-per-pass parity test, the batched swap's
# kernel trace in the default and the batch-invariant mode, the per-rank pass-size probe.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_schedule.py -m gpu -q -s -k "timed_pass_size or call_surface" > gpurun_out/r05a_tests.log 2>&1; tail -12 gpurun_out/r05a_tests.log
bash tools/prof_swap.sh r05a stats det
timeout 600 python tools/probes/rank_pass_sizes.py > gpurun_out/r05a_pass_sizes.log 2>&1; tail -5 gpurun_out/r05a_pass_sizes.log
