# GPU call r05b: batch-invariant plans as the default (virtual split-K): the whole GPU suite, the batched swap's kernel trace in
# both plan modes, the generator line and the single swap in both modes.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x --durations=8 > gpurun_out/r05b_tests.log 2>&1; tail -15 gpurun_out/r05b_tests.log
bash tools/prof_swap.sh r05b stats nodet
for d in 1 0; do
  HAIRFAST_DETERMINISTIC=$d python bench.py --no-cpu-baseline --no-exact-f32 --swap-triples 0 --steps 20 --warmup 3 > gpurun_out/r05b_gen_det$d.json 2> gpurun_out/r05b_gen_det$d.err
  HAIRFAST_DETERMINISTIC=$d python bench.py --workload swap256 --triples 12 --swap-batch 1 --warmup 2 --no-kernel-events --no-verify > gpurun_out/r05b_single_det$d.json 2> gpurun_out/r05b_single_det$d.err
  python -c "
import json
g=json.load(open('gpurun_out/r05b_gen_det$d.json')); s=json.load(open('gpurun_out/r05b_single_det$d.json'))
print('det=$d generator img/s', g['value'], 'single swap ms', s['ms_per_step'])"
done
