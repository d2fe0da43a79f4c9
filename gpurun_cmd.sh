set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -40 > gpurun_out/r02a_tests.log
python bench.py --steps 10 --warmup 2 > gpurun_out/r02a_bench.log 2>gpurun_out/r02a_bench.err
HF_FORCE_DIST=1 MASTER_PORT=29701 python bench.py --workload swap256 --triples 32 --warmup 2 > gpurun_out/r02a_swap32.log 2>gpurun_out/r02a_swap32.err
python tools/bench_encoders.py > gpurun_out/r02a_encoders.log 2>&1
tail -5 gpurun_out/r02a_tests.log
