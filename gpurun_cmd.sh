# GPU call r06bi: default bench line at the final commit (the driver's invocation) + the stats pass for the launch census
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py > gpurun_out/r06_bench.json 2> gpurun_out/r06_bench.err; echo "bench rc=$?"
tail -c 600 gpurun_out/r06_bench.json
