# GPU call r06bl: patterns of the interleaved fragment reads: one behind each of the first 8 MFMAs (hip), behind the last 8 (ilv2), two behind each of the first 4 (ilv3)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
C=$GRAFT_REPO_ROOT/hairfastgan_amd/csrc
for v in hip ilv2 ilv3 hip ilv2 ilv3 hip ilv2 ilv3; do echo "== $v"; HAIRFAST_HIP_LIB=$C/libhairfast_$v.so python bench.py --no-cpu-baseline --no-exact-f32 --swap-triples 0 --steps 40 --warmup 5 --no-kernel-events | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06bl_ilv_patterns.txt
