# GPU call r06be: the whole GPU suite, then rocprofv3 stats + PMC passes of both workloads and the default bench line at the final commit (tag r06)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
bash tools/profile_all.sh r06 2>&1 | grep "rc="
python bench.py > gpurun_out/r06_bench.json 2> gpurun_out/r06_bench.err; echo "bench rc=$?"
tail -c 700 gpurun_out/r06_bench.json
