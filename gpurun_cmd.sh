# GPU call r06ac: 128-channel GEMM blocks in the batched swap (tuning 2 = off) + GPU tests of the GEMM users
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_encoders.py tests/test_gpu_sean.py tests/test_gpu_clip.py -x -q -m gpu 2>&1 | tail -3
for t in 2 0 2 0; do echo "== tuning $t"; python tools/probes/bench_tuned.py $t --workload swap256 --triples 64 --swap-batch 32 --warmup 1 --no-kernel-events --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['verified']['equal'])"; done
