cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parsing.py -q -x -k "shape_adaptor" 2>&1 | tail -3
python tools/probes/shape_adaptor_launches.py 2>&1 | grep -v "amdgpu\|Warn\|warn" | grep -A8 "total device"
python bench.py --workload swap256 --triples 8 --warmup 1 --swap-batch 1 --no-kernel-events 2>/dev/null | head -c 200; echo
