cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_schedule.py -q -x 2>&1 | tail -4
PROBE_T=1,1,2 python tools/probes/swap_batch_sizes.py 2>&1 | grep -v amdgpu | tail -4
HAIRFAST_EMBED_OVERLAP=0 PROBE_T=1,1,2 python tools/probes/swap_batch_sizes.py 2>&1 | grep -v amdgpu | tail -4
python bench.py --workload swap256 --triples 16 --warmup 1 --swap-batch 1 --no-kernel-events 2>/dev/null | head -c 200; echo
