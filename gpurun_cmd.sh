cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -5 | tee gpurun_out/r02g_tests.log
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r02g_swap -o bench -- python $R/bench.py --workload swap256 --triples 16 --warmup 1 --no-kernel-events > $R/gpurun_out/prof_r02g_swap.log 2>&1
cd $R
python bench.py --steps 30 --warmup 5 > gpurun_out/r02g_bench.json 2> gpurun_out/r02g_bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02g_bench.json'))
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['achieved'], d['roofline']['frac'], d.get('exact_f32',{}).get('value'), d.get('f16_mode',{}).get('value'), d.get('swap_schedule',{}).get('value'), d.get('swap_pipeline'))
PY
