cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parsing.py tests/test_gpu_encoders.py -q -x 2>&1 | tail -3
python tools/probes/shape_adaptor_launches.py 2>&1 | grep -v "amdgpu\|Warn\|warn" | grep -A4 "total device"
python tools/bench_encoders.py 2>&1 | grep -v amdgpu | grep "B=3"
python bench.py --workload swap256 --triples 32 --warmup 1 --no-kernel-events 2>/dev/null | head -c 200; echo
