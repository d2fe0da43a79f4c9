# GPU call r06az: row pipeline finishing both rows of a wave in one pass (fin1) vs one pass per row (hip)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
C=$GRAFT_REPO_ROOT/hairfastgan_amd/csrc
HAIRFAST_HIP_LIB=$C/libhairfast_fin1.so python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "conv_rows or generator1024" 2>&1 | tail -2
for v in hip fin1 hip fin1; do echo "== $v"; HAIRFAST_HIP_LIB=$C/libhairfast_$v.so python bench.py --no-cpu-baseline --no-exact-f32 --swap-triples 0 --steps 40 --warmup 5 --no-kernel-events | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06az_fin1_bench.txt
