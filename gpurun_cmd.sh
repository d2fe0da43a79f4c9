cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -6 | tee gpurun_out/r02v_tests.log
python bench.py --steps 30 --warmup 5 > gpurun_out/r02v_bench.json 2> gpurun_out/r02v_bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02v_bench.json'))
print({k:d[k] for k in ('value','ms_per_step')}, d.get('roofline'), d.get('exact_f32'), d.get('f16_mode'), d.get('swap_schedule',{}).get('value'), d.get('swap_pipeline'))
PY
