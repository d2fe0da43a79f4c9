# Round-end check on the GPU box (run through `gpurun -- bash gpurun_cmd.sh` from the repo root):
# GPU parity suite, smoke, the default bench line and the batched-swap workload.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -4 | tee gpurun_out/tests_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; head -c 400 gpurun_out/bench.json; echo
python bench.py --workload swap256 --triples 32 --warmup 1 --no-kernel-events 2> gpurun_out/swap256.err | head -c 300; echo
