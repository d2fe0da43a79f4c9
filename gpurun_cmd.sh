# GPU call r05m: gemm1x1_h at two blocks per CU (launch bound 2 waves / SIMD) A/B per shape and on the batched + single swap
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/probes/gemm_shapes.py > gpurun_out/r05m_gemm_2.log 2>&1
HAIRFAST_HIP_LIB=$GRAFT_REPO_ROOT/hairfastgan_amd/csrc/libhairfast_gemm1.so python tools/probes/gemm_shapes.py > gpurun_out/r05m_gemm_1.log 2>&1
paste -d'\n' gpurun_out/r05m_gemm_2.log gpurun_out/r05m_gemm_1.log | grep -v amdgpu.ids
for v in hip gemm1; do
  HAIRFAST_HIP_LIB=$GRAFT_REPO_ROOT/hairfastgan_amd/csrc/libhairfast_$v.so python bench.py --workload swap256 --triples 64 --swap-batch 32 --warmup 1 --no-kernel-events > gpurun_out/r05m_swap_$v.json 2> gpurun_out/r05m_swap_$v.err
  HAIRFAST_HIP_LIB=$GRAFT_REPO_ROOT/hairfastgan_amd/csrc/libhairfast_$v.so python bench.py --workload swap256 --triples 12 --swap-batch 1 --warmup 2 --no-kernel-events --no-verify > gpurun_out/r05m_single_$v.json 2> gpurun_out/r05m_single_$v.err
  python -c "
import json; d=json.load(open('gpurun_out/r05m_swap_$v.json')); s=json.load(open('gpurun_out/r05m_single_$v.json')); print('$v', d['value'], 'triples/s', d['verified']['equal'], 'single ms', s['ms_per_step'])"
done
python -m pytest tests/test_gpu_parity.py -m gpu -q -k "block_order" 2>&1 | tail -2
