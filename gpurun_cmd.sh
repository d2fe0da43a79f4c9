# GPU call r05t: s2mt with the chunk's weight fragments in registers (A/B vs -DHF_ENC_S2MT_AREG=0), batched stage glue; swap A/B; tests
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/probes/enc_layers_pre.py 2>&1 | grep -E "s2|lib" > gpurun_out/r05t_layers_areg1.log
HAIRFAST_HIP_LIB=$GRAFT_REPO_ROOT/hairfastgan_amd/csrc/libhairfast_areg0.so python tools/probes/enc_layers_pre.py 2>&1 | grep -E "s2|lib" > gpurun_out/r05t_layers_areg0.log
paste -d'\n' gpurun_out/r05t_layers_areg1.log gpurun_out/r05t_layers_areg0.log
for v in hip areg0 hip areg0; do
  HAIRFAST_HIP_LIB=$GRAFT_REPO_ROOT/hairfastgan_amd/csrc/libhairfast_$v.so python bench.py --workload swap256 --triples 64 --swap-batch 32 --warmup 1 --no-kernel-events > gpurun_out/r05t_swap_$v.json 2> gpurun_out/r05t_swap_$v.err
  python -c "
import json; d=json.load(open('gpurun_out/r05t_swap_$v.json')); print('$v', d['value'], 'triples/s', d['verified']['equal'])"
done
python bench.py --workload swap256 --triples 4 --swap-batch 1 --warmup 1 --no-kernel-events 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('single', d['ms_per_step'], d['value'])"
python -m pytest tests/test_gpu_schedule.py tests/test_gpu_pipeline.py tests/test_gpu_encoders.py -m gpu -q -x 2>&1 | tail -3
