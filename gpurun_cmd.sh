cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > gpurun_out/r02h_bench.json 2> gpurun_out/r02h_bench.err; python -c "
import json; d=json.load(open('gpurun_out/r02h_bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic_source'][:40], d['cpu_baseline']['value'], d['swap_pipeline']['value'], d['swap_pipeline']['single_swap'])"
