cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_encoders.py tests/test_gpu_schedule.py -q -x -k "latent or hairfast or swap" 2>&1 | tail -5
python bench.py --workload swap256 --triples 32 --warmup 1 --no-kernel-events 2>/dev/null | head -c 200; echo
python bench.py --workload swap256 --triples 16 --warmup 1 --swap-batch 1 --no-kernel-events 2>/dev/null | head -c 200; echo
