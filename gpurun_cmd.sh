set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/r02g_tests.log
python bench.py --steps 20 --warmup 3 > gpurun_out/r02g_bench.log 2>gpurun_out/r02g_bench.err
python tools/bench_batch.py > gpurun_out/r02g_batch.log 2>&1
tail -5 gpurun_out/r02g_tests.log; cat gpurun_out/r02g_batch.log
python - <<'PY'
import json
l=[x for x in open('gpurun_out/r02g_bench.log') if x.startswith('{"metric')][-1]
d=json.loads(l)
print(d['value'], d['ms_per_step'], d.get('exact_f32',{}).get('value'), d.get('f16_mode',{}).get('value'))
print(d.get('swap_schedule',{}).get('ms_per_triple'), d.get('swap_pipeline'))
print(d['cpu_baseline'])
for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['share_of_timed_region']): print(k, v)
print(d['roofline']); print(d.get('roofline_hbm'))
PY
