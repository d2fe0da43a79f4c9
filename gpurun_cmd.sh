cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -q -k "fuse or blur or up or range" 2>&1 | tail -4
PROBE_TUNE=0 python tools/probes/gen_layers.py 2>&1 | grep -v amdgpu > /dev/null
PROBE_TUNE=0 python tools/probes/gen_layers.py 2>&1 | grep -v amdgpu | tee gpurun_out/r02q_layers.log
