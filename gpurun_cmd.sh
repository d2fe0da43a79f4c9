# GPU call r06bd: 8^2 same-resolution layer on the tap GEMM from the canonical batch (A/B by HAIRFAST... git stash is not available on the box: two runs, rule on = HEAD)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_schedule.py -x -q -m gpu 2>&1 | tail -3
for v in 1 2 3; do python bench.py --no-cpu-baseline --no-exact-f32 --swap-triples 0 --steps 40 --warmup 5 --no-kernel-events | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06bd_bench.txt
sed -i 's/ or (h \* w == 64 and n >= 192 and n <= 2048)$//' hairfastgan_amd/_marshal.py
echo "== rule off"
for v in 1 2 3; do python bench.py --no-cpu-baseline --no-exact-f32 --swap-triples 0 --steps 40 --warmup 5 --no-kernel-events | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"; done 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r06bd_bench.txt
