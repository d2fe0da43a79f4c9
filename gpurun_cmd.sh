# GPU call r05r: stride-2 multi-tile form with the rounds rule (x4 / x2 / one tile) A/B per layer and on the batched swap; GPU tests
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/probes/enc_layers_pre.py 2>&1 | grep -E "s2|lib" > gpurun_out/r05r_layers_mt.log
HAIRFAST_HIP_LIB=$GRAFT_REPO_ROOT/hairfastgan_amd/csrc/libhairfast_mt1.so python tools/probes/enc_layers_pre.py 2>&1 | grep -E "s2|lib" > gpurun_out/r05r_layers_mt1.log
paste -d'\n' gpurun_out/r05r_layers_mt.log gpurun_out/r05r_layers_mt1.log
for v in hip mt1 hip mt1; do
  HAIRFAST_HIP_LIB=$GRAFT_REPO_ROOT/hairfastgan_amd/csrc/libhairfast_$v.so python bench.py --workload swap256 --triples 64 --swap-batch 32 --warmup 1 --no-kernel-events > gpurun_out/r05r_swap_$v.json 2> gpurun_out/r05r_swap_$v.err
  python -c "
import json; d=json.load(open('gpurun_out/r05r_swap_$v.json')); print('$v', d['value'], 'triples/s', d['verified']['equal'])"
done
python -m pytest tests -m gpu -q -x 2>&1 | tail -3
