cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_schedule.py -q -x 2>&1 | tail -15
for b in 1 2 4 8; do
  python bench.py --workload swap256 --triples 32 --warmup 1 --swap-batch $b --no-kernel-events 2>gpurun_out/r02r_swap_b$b.err | head -c 420; echo
done | tee gpurun_out/r02r_swapbatch.log
