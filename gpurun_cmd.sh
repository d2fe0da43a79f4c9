cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/bench_batch.py 2>&1 | grep -v amdgpu | tee gpurun_out/r02z_batch_scaling.txt
PROBE_BATCH=1 python tools/probes/forward_launches.py 2>&1 | grep -v "amdgpu\|Warn\|warn" | head -24
