cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_schedule.py tests/test_gpu_encoders.py -m gpu -q -k "invariant or graphed or oracle" > gpurun_out/t3.log 2>&1; tail -25 gpurun_out/t3.log
python tools/probes/swap_ops.py 1 > gpurun_out/swap_ops_1.txt 2>&1; tail -5 gpurun_out/swap_ops_1.txt
export HAIRFAST_HIP_LIB=$GRAFT_REPO_ROOT/hairfastgan_amd/csrc/libhairfast_trace.so
for cfg in "64 32 512 8 fuse" "128 64 256 8 fuse" "256 128 128 8 fuse" "512 512 64 8 pre"; do
  echo "=== $cfg" >> gpurun_out/trace3.txt
  python tools/probes/trace_layer.py $cfg >> gpurun_out/trace3.txt 2>&1
done
tail -5 gpurun_out/trace3.txt
