cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B="python bench.py --no-cpu-baseline --no-exact-f32 --swap-triples 0"
p='import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d["value"], d["ms_per_step"], d.get("priming_steps"), d["roofline"]["launches"], d["roofline"]["frac"])'
for r in 1 2; do
echo "== default"; $B | python -c "$p"
echo "== default --priming 0 --event-every 5"; $B --priming 0 --event-every 5 | python -c "$p"
echo "== 20 steps 3 warm-up"; $B --steps 20 --warmup 3 | python -c "$p"
echo "== 40 steps, no events"; $B --no-kernel-events --steps 40 --warmup 5 | python -c "$p"
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06bh_bench_modes.txt
