cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -q 2>&1 | tail -4
python tools/probes/forward_launches.py 2>&1 | grep -v "amdgpu\|Warn\|warn" | head -12
python bench.py --steps 30 --warmup 5 --swap-triples 0 2>/dev/null | head -c 250
