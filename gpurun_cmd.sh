# GPU call r06bn: row pipeline as a ping-pong (hip: + interleaved reads; rppn: without) vs the one-phase form (rpp0)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
C=$GRAFT_REPO_ROOT/hairfastgan_amd/csrc
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "conv_rows or generator1024" 2>&1 | tail -2
for v in rpp0 hip rppn rpp0 hip rppn; do echo "== $v"; HAIRFAST_HIP_LIB=$C/libhairfast_$v.so python bench.py --no-cpu-baseline --no-exact-f32 --swap-triples 0 --steps 40 --warmup 5 --no-kernel-events | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06bn_rows_pingpong.txt
for v in rpp0 hip rppn; do echo "== $v"; HAIRFAST_HIP_LIB=$C/libhairfast_$v.so PROBE_TUNE=0 python tools/probes/gen_layers.py 2>&1 | grep "same  32"; done | tee -a gpurun_out/r06bn_rows_pingpong.txt
