(timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -5)
python tools/bench_layers.py --batch 8 --iters 5 --only none 2>&1 | tail -9
(timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-400)
