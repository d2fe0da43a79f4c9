# GPU call r06ao: noise maps from 64^2 on a side stream that waits for the current stream's position; style kernels with the weight row kept for 8 batch elements
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in 0 1 0 1; do echo "== HAIRFAST_NOISE_STREAM=$v"; HAIRFAST_NOISE_STREAM=$v python bench.py --no-cpu-baseline --no-exact-f32 --swap-triples 0 --steps 40 --warmup 5 --no-kernel-events | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06ao_noise_stream.txt
python tools/probes/forward_launches.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06ao_forward_launches.txt | head -40
