mkdir -p gpurun_out
(timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -6)
(timeout 900 python bench.py --steps 10 --warmup 3 2>&1 | tail -1) > gpurun_out/bench_r01j.log; cat gpurun_out/bench_r01j.log
python tools/bench_encoders.py 2>&1 | tail -4
