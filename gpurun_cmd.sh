(timeout 1200 python -m pytest tests/test_gpu_schedule.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -12)
(timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1) > gpurun_out/bench_r01n.log; python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_r01n.log').read())
print(d['value'], d['ms_per_step'], json.dumps(d.get('swap_schedule'))[:600])
PY
