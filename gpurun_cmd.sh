# GPU call r06av: HEAD with interleaved fragment reads in both ping-pong K loops: parity of the generator and encoder kernels; batched swap A/B encoder interleave on (hip) / off (encilv0)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
C=$GRAFT_REPO_ROOT/hairfastgan_amd/csrc
python -m pytest tests/test_gpu_parity.py tests/test_gpu_encoders.py -x -q -m gpu 2>&1 | tail -2
for v in encilv0 hip encilv0 hip; do echo "== $v"; HAIRFAST_HIP_LIB=$C/libhairfast_$v.so python bench.py --workload swap256 --triples 64 --swap-batch 32 --warmup 1 --no-kernel-events --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d.get('value'), d.get('ms_per_step'), d.get('verified'))"; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06av_swap_enc_ilv.txt
