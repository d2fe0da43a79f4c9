# GPU call r06j: ping-pong copy roles (activations by the half that idles first) on the fused and same-resolution kernels + parity
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
PROBE_REPS=2 python tools/probes/fuse_ab.py r0 hip > gpurun_out/r06j_roles.txt 2>&1
cat gpurun_out/r06j_roles.txt
C=$GRAFT_REPO_ROOT/hairfastgan_amd/csrc
for v in r0 hip r0 hip; do echo "== $v"; HAIRFAST_HIP_LIB=$C/libhairfast_$v.so PROBE_TUNE=0 python tools/probes/gen_layers.py 2>&1 | grep -v amdgpu | grep "same\|upfu\|512->256\|512->512" ; done > gpurun_out/r06j_gen_layers.txt
cat gpurun_out/r06j_gen_layers.txt
python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
