# GPU call r05g (round check): GPU parity suite, smoke, the default bench line, rocprofv3 stats + PMC passes of the generator
# workload and of the batched swap at its timed pass size.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --durations=5 > gpurun_out/tests_gpu.log 2>&1; tail -6 gpurun_out/tests_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; python -c "
import json; d=json.load(open('gpurun_out/bench.json')); p=d['swap_pipeline']
print(d['value'], d['f16_mode']['value'], p['value'], p['single_swap']['ms_per_swap'], p['single_swap_graph'], p.get('verified'))"
bash tools/profile_bench.sh r05a
bash tools/prof_swap.sh r05a stats pmc
