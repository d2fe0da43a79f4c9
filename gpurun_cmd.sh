# GPU call r06ay: rocprofv3 stats + PMC passes of both workloads at HEAD (tag r06), clock / power telemetry, default bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/profile_all.sh r06 2>&1 | tail -12
python bench.py > gpurun_out/r06_bench.json 2> gpurun_out/r06_bench.err; echo "bench rc=$?"
tail -c 700 gpurun_out/r06_bench.json
