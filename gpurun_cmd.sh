(timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -3)
python tools/bench_layers.py --batch 8 --iters 5 --only none 2>&1 | tail -9 | grep blur
(timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --swap-triples 0 2>&1 | tail -1 | cut -c1-200)
