# GPU call r06v: small-plane upsampling in two launches, second form (tap planes staged in LDS)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "two_launches or small_modules or all_ranges" 2>&1 | tail -3
for f in 0 1 0 1; do echo "== HAIRFAST_SMALL_UP_FUSED=$f"; HAIRFAST_SMALL_UP_FUSED=$f python bench.py --no-cpu-baseline --no-exact-f32 --swap-triples 0 --steps 40 --warmup 5 --no-kernel-events | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"; done
python tools/probes/forward_launches.py 2>&1 | grep "small_up_blur\|small_combine\|blur4x4\|total device"
