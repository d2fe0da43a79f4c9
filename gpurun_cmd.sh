# GPU call r05x: rows epilogue specialised on its per-launch switches (no per-element branches, pointer-walked channel rows) + the fast
# prologue of conv_enc_h: A/B per layer, GEMM shapes, block timeline, batched swap
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
C=$GRAFT_REPO_ROOT/hairfastgan_amd/csrc
for v in base hip; do
  HAIRFAST_HIP_LIB=$C/libhairfast_$v.so python tools/probes/enc_layers_pre.py 2>&1 | grep -v "amdgpu.ids\|lib:" | cut -c1-21,41-75 > gpurun_out/r05x_layers_$v.log
  HAIRFAST_HIP_LIB=$C/libhairfast_$v.so python tools/probes/gemm_shapes.py 2>&1 | grep -v "amdgpu.ids\|lib:" | cut -c1-31,44-90 > gpurun_out/r05x_gemm_$v.log
done
echo "layers: base | new"
paste -d'|' gpurun_out/r05x_layers_base.log gpurun_out/r05x_layers_hip.log | cut -c1-56,78-112
paste -d'|' gpurun_out/r05x_gemm_base.log gpurun_out/r05x_gemm_hip.log | cut -c1-78,110-160
HAIRFAST_HIP_LIB=$C/libhairfast_enctrace.so python tools/probes/trace_enc_layer.py 96 64 64 128 128 2>&1 | grep -v amdgpu.ids > gpurun_out/r05x_trace_enc.txt
cat gpurun_out/r05x_trace_enc.txt
for v in base hip base hip; do
  HAIRFAST_HIP_LIB=$C/libhairfast_$v.so python bench.py --workload swap256 --triples 64 --swap-batch 32 --warmup 1 --no-kernel-events > gpurun_out/r05x_swap_$v.json 2> gpurun_out/r05x_swap_$v.err
  python -c "
import json; d=json.load(open('gpurun_out/r05x_swap_$v.json')); print('$v', d['value'], 'triples/s', d['verified']['equal'])"
done
