# GPU call r06n: full GPU test suite + bench line at the current state
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | tail -5
python bench.py > gpurun_out/r06n_bench.json 2> gpurun_out/r06n_bench.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/r06n_bench.json'))
print({k: d.get(k) for k in ['value', 'ms_per_step']})
print('fam', {k: (v['avg_launch_ms'], v['frac']) for k, v in d['roofline_families'].items()} if 'roofline_families' in d else None)
sp = d.get('swap_pipeline') or {}
print('swap', sp.get('value'), sp.get('single_swap'), {k: sp.get(k) for k in ['ms_per_triple']})
print('f16', (d.get('f16_mode') or {}).get('value'), 'f32', (d.get('exact_f32') or {}).get('value'))
PY
