# GPU call r05h: unit -> unit hand-off (HAIRFAST_UNIT_CHAIN) A/B on the batched and the single swap, then the whole GPU suite.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for c in 1 0; do
  HAIRFAST_UNIT_CHAIN=$c python bench.py --workload swap256 --triples 64 --swap-batch 32 --warmup 1 --no-kernel-events > gpurun_out/r05h_swap_chain$c.json 2> gpurun_out/r05h_swap_chain$c.err
  HAIRFAST_UNIT_CHAIN=$c python bench.py --workload swap256 --triples 12 --swap-batch 1 --warmup 2 --no-kernel-events --no-verify > gpurun_out/r05h_single_chain$c.json 2> gpurun_out/r05h_single_chain$c.err
  python -c "
import json
a=json.load(open('gpurun_out/r05h_swap_chain$c.json')); s=json.load(open('gpurun_out/r05h_single_chain$c.json'))
print('chain=$c batched', a['value'], 'triples/s', a['verified']['equal'], 'single swap ms', s['ms_per_step'])"
done
python -m pytest tests -m gpu -q > gpurun_out/r05h_tests.log 2>&1; tail -5 gpurun_out/r05h_tests.log
