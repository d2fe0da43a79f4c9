# GPU call r06ag: same-resolution kernel instantiated for the split + ToRGB-slab epilogue (hip) vs run-time flags (epi0)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
C=$GRAFT_REPO_ROOT/hairfastgan_amd/csrc
for v in epi0 hip epi0 hip; do echo "== $v"; HAIRFAST_HIP_LIB=$C/libhairfast_$v.so python bench.py --no-cpu-baseline --no-exact-f32 --swap-triples 0 --steps 40 --warmup 5 --no-kernel-events | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"; done
python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
