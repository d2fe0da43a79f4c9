# GPU call r05n: split-K plan by the one-block-per-CU cost model A/B on the single and the batched swap; encoder GPU tests
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in hip planold; do
  HAIRFAST_HIP_LIB=$GRAFT_REPO_ROOT/hairfastgan_amd/csrc/libhairfast_$v.so python bench.py --workload swap256 --triples 24 --swap-batch 1 --warmup 3 --no-kernel-events --no-verify > gpurun_out/r05n_single_$v.json 2> gpurun_out/r05n_single_$v.err
  HAIRFAST_HIP_LIB=$GRAFT_REPO_ROOT/hairfastgan_amd/csrc/libhairfast_$v.so python bench.py --workload swap256 --triples 64 --swap-batch 32 --warmup 1 --no-kernel-events > gpurun_out/r05n_swap_$v.json 2> gpurun_out/r05n_swap_$v.err
  python -c "
import json; d=json.load(open('gpurun_out/r05n_swap_$v.json')); s=json.load(open('gpurun_out/r05n_single_$v.json')); print('$v', d['value'], 'triples/s', d['verified']['equal'], 'single ms', s['ms_per_step'])"
done
HAIRFAST_HIP_LIB=$GRAFT_REPO_ROOT/hairfastgan_amd/csrc/libhairfast_planold.so python bench.py --workload swap256 --triples 24 --swap-batch 1 --warmup 3 --no-kernel-events --no-verify 2>/dev/null | python -c "import json,sys; print('planold again single ms', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"
python bench.py --workload swap256 --triples 24 --swap-batch 1 --warmup 3 --no-kernel-events --no-verify 2>/dev/null | python -c "import json,sys; print('new again single ms', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"
python -m pytest tests/test_gpu_encoders.py tests/test_gpu_schedule.py -m gpu -q 2>&1 | tail -3
