# GPU call r06x: noise draw on a side stream A/B + full GPU suite
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for f in 0 1 0 1; do echo "== HAIRFAST_NOISE_SIDE_STREAM=$f"; HAIRFAST_NOISE_SIDE_STREAM=$f python bench.py --no-cpu-baseline --no-exact-f32 --swap-triples 0 --steps 60 --warmup 5 --no-kernel-events | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"; done
python -m pytest tests -x -q -m gpu 2>&1 | tail -4
