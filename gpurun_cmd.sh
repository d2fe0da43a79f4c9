# GPU call r05z: final-state evidence of round 5 - full GPU suite, smoke, the default bench line, rocprofv3 stats + PMC of the batched swap
# (cut to its timed region), rocprofv3 stats of the generator workload
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 700 python -m pytest tests -m gpu -q -x > gpurun_out/r05z_tests.log 2>&1; tail -3 gpurun_out/r05z_tests.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 600 python bench.py > gpurun_out/r05z_bench.json 2> gpurun_out/r05z_bench.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/r05z_bench.json'))
sp = d.get('swap_pipeline', {})
print('generator', d['value'], 'img/s', d['ms_per_step'], 'ms; roofline', d['roofline'].get('kernel'), d['roofline'].get('frac'), '; f16', d.get('f16_mode', {}).get('value'))
print('swap', sp.get('value'), 'triples/s verified', sp.get('verified', {}).get('equal'), 'single', sp.get('single_swap', {}).get('ms_per_swap'), 'graphed', sp.get('single_swap_graph', {}).get('ms_per_swap'))
PY
echo "t=$SECONDS"
bash tools/prof_swap.sh r05z stats pmc
echo "t=$SECONDS"
if [ $SECONDS -lt 720 ]; then
  cd /tmp && export TMPDIR=/tmp
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r05z_stats -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-exact-f32 --swap-triples 0 --steps 3 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/prof_r05z_stats.log 2>&1
  echo "gen stats rc=$? t=$SECONDS"
fi
