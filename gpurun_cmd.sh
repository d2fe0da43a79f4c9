# GPU call r06ba: row pipeline with the second wave of each SIMD running its epilogue arithmetic after its MFMAs (stag) vs both at the head of the next step (hip)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
C=$GRAFT_REPO_ROOT/hairfastgan_amd/csrc
HAIRFAST_HIP_LIB=$C/libhairfast_stag.so python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "conv_rows or generator1024" 2>&1 | tail -2
for v in hip stag hip stag; do echo "== $v"; HAIRFAST_HIP_LIB=$C/libhairfast_$v.so python bench.py --no-cpu-baseline --no-exact-f32 --swap-triples 0 --steps 40 --warmup 5 --no-kernel-events | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06ba_stag_bench.txt
