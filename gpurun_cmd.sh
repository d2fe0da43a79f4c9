# GPU call r05zb: nontemporal stores in the encoder conv's epilogue (-DHF_NT_STORES in convh_enc.hip) on the batched swap
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
C=$GRAFT_REPO_ROOT/hairfastgan_amd/csrc
for v in hip nt hip nt; do
  HAIRFAST_HIP_LIB=$C/libhairfast_$v.so python bench.py --workload swap256 --triples 64 --swap-batch 32 --warmup 1 --no-kernel-events > gpurun_out/r05zb_swap_$v.json 2> gpurun_out/r05zb_swap_$v.err
  python -c "
import json; d=json.load(open('gpurun_out/r05zb_swap_$v.json')); print('$v', d['value'], 'triples/s', d['verified']['equal'])"
done
