# GPU call r06bc: ping-pong K loop, first half issues 1 / 2 of its activation copies under the latency of its first fragments (ex1 / ex2) vs all in its idle phase (hip)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
C=$GRAFT_REPO_ROOT/hairfastgan_amd/csrc
for v in ex1 ex2; do HAIRFAST_HIP_LIB=$C/libhairfast_$v.so python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "generator1024 or modconv" 2>&1 | tail -1; done
for v in hip ex1 ex2 hip ex1 ex2; do echo "== $v"; HAIRFAST_HIP_LIB=$C/libhairfast_$v.so python bench.py --no-cpu-baseline --no-exact-f32 --swap-triples 0 --steps 40 --warmup 5 --no-kernel-events | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06bc_early_x_bench.txt
