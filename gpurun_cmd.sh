# GPU call r05j: the generator kernels' block order (cout-tiles-fastest where it moves fewer bytes from beyond L2) A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
PROBE_TUNE=0 python tools/probes/gen_layers.py > gpurun_out/r05j_gen_swap.log 2>&1
HAIRFAST_HIP_LIB=$GRAFT_REPO_ROOT/hairfastgan_amd/csrc/libhairfast_noswap.so PROBE_TUNE=0 python tools/probes/gen_layers.py > gpurun_out/r05j_gen_noswap.log 2>&1
paste -d'\n' gpurun_out/r05j_gen_swap.log gpurun_out/r05j_gen_noswap.log | grep -v amdgpu.ids
for v in hip noswap; do
  HAIRFAST_HIP_LIB=$GRAFT_REPO_ROOT/hairfastgan_amd/csrc/libhairfast_$v.so python bench.py --no-cpu-baseline --no-exact-f32 --swap-triples 0 --steps 20 --warmup 3 > gpurun_out/r05j_gen_$v.json 2> gpurun_out/r05j_gen_$v.err
  python -c "
import json; d=json.load(open('gpurun_out/r05j_gen_$v.json')); print('$v', d['value'], 'img/s')"
done
python -m pytest tests/test_gpu_parity.py -m gpu -q > gpurun_out/r05j_tests.log 2>&1; tail -3 gpurun_out/r05j_tests.log
