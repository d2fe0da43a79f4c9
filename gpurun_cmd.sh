# GPU call r06ar: last StyledConv finishes ToRGB in its epilogue (ABI 13): full parity + A/B against the two-launch path (tuning... HAIRFAST_IMAGE_FUSE=0)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
for v in 0 1 0 1; do echo "== HAIRFAST_IMAGE_FUSE=$v"; HAIRFAST_IMAGE_FUSE=$v python bench.py --no-cpu-baseline --no-exact-f32 --swap-triples 0 --steps 40 --warmup 5 --no-kernel-events | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06ar_bench.txt
