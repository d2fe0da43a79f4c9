cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/probes/fuse_layers.py > gpurun_out/r02f_fuse_layers.log 2>&1
cat gpurun_out/r02f_fuse_layers.log
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fused_blur" 2>&1 | tail -3
