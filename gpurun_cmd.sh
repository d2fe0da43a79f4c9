cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -4
python bench.py --workload swap256 --triples 32 --warmup 1 --no-kernel-events 2>/dev/null | head -c 200; echo
python bench.py --workload swap256 --triples 8 --warmup 1 --swap-batch 1 --no-kernel-events 2>/dev/null | head -c 200; echo
python tools/bench_encoders.py 2>&1 | grep -v amdgpu | tail -12
