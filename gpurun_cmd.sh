# GPU call r06ad: 128-channel GEMM blocks for the 128-pixel tile form too (heads patch GEMM)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
C=$GRAFT_REPO_ROOT/hairfastgan_amd/csrc
for v in hip g128c hip g128c; do HAIRFAST_HIP_LIB=$C/libhairfast_$v.so python tools/probes/gemm_shapes.py 2>&1 | grep -v amdgpu | grep "lib\|heads\|SEAN 512\|CLIP"; done
