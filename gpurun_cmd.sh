cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_encoders.py tests/test_gpu_schedule.py tests/test_gpu_parsing.py -q -x 2>&1 | tail -3
for b in 8; do
  python bench.py --workload swap256 --triples 32 --warmup 1 --swap-batch $b --no-kernel-events 2>gpurun_out/r02u_swap_b$b.err | head -c 330; echo
done | tee gpurun_out/r02u_swapbatch.log
