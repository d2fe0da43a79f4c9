# Round check on the GPU box (run through `gpurun -- bash gpurun_cmd.sh` from the repo root): GPU parity suite, smoke, the default
# bench line, rocprofv3 stats + PMC passes of the generator workload and of the batched swap at its timed pass size, the swap
# workload once more with RCCL initialised at world 1.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --durations=5 > gpurun_out/tests_gpu.log 2>&1; tail -4 gpurun_out/tests_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; python -c "
import json; d=json.load(open('gpurun_out/bench.json')); p=d['swap_pipeline']
print(d['value'], d['f16_mode']['value'], p['value'], p['single_swap']['ms_per_swap'], p['single_swap_graph'], p.get('verified',{}).get('equal'))"
bash tools/profile_bench.sh r05c
bash tools/prof_swap.sh r05c stats pmc
HF_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29671 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 python bench.py --workload swap256 --triples 64 --no-kernel-events 2> gpurun_out/bench_dist.err | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('forced RCCL world 1:', d['value'], 'triples/s', d['config']['gather'], d['verified']['equal'], d['balance'])"
