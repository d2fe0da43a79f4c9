cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_encoders.py tests/test_gpu_parsing.py tests/test_gpu_schedule.py -q -x 2>&1 | tail -3
python bench.py --workload swap256 --triples 32 --warmup 1 --no-kernel-events 2>/dev/null | head -c 200; echo
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r02f_swap -o bench -- python $R/bench.py --workload swap256 --triples 16 --warmup 1 --no-kernel-events > $R/gpurun_out/prof_r02f_swap.log 2>&1
