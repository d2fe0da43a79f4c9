set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parsing.py -m gpu -x -q -s 2>&1 | tail -25 > gpurun_out/r02h_parsing.log
python -m pytest tests -m gpu -q 2>&1 | tail -12 > gpurun_out/r02h_tests.log
HF_FORCE_DIST=1 MASTER_PORT=29721 python bench.py --workload swap256 --triples 32 --warmup 2 > gpurun_out/r02h_swap32.log 2>gpurun_out/r02h_swap32.err
cat gpurun_out/r02h_parsing.log; tail -6 gpurun_out/r02h_tests.log; head -c 400 gpurun_out/r02h_swap32.log; tail -2 gpurun_out/r02h_swap32.err
