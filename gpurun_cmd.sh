set -x
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/r02l_tests.log
HF_FORCE_DIST=1 MASTER_PORT=29731 python bench.py --workload swap256 --triples 32 --warmup 2 > gpurun_out/r02l_swap32.log 2>gpurun_out/r02l_swap32.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r02b_swap -o bench -- python $R/bench.py --workload swap256 --triples 6 --warmup 1 --no-kernel-events > $R/gpurun_out/prof_r02b_swap.log 2>&1
cd $R
tail -5 gpurun_out/r02l_tests.log; head -c 330 gpurun_out/r02l_swap32.log; echo; head -14 gpurun_out/prof_r02b_swap/bench_kernel_stats.csv | cut -c1-150
