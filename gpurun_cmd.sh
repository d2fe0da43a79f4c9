# GPU call r06ah: final profiles (rocprofv3 kernel trace + PMC passes of both workloads) and the full bench line at HEAD
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/profile_all.sh r06
python bench.py > gpurun_out/r06_bench.json 2> gpurun_out/r06_bench.err
tail -c 700 gpurun_out/r06_bench.json
