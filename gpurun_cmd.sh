# GPU call r06bm: ping-pong K loop, a half passes the barrier its partner waits at BEFORE the MFMAs of its last tap (eb) vs behind them (hip)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
C=$GRAFT_REPO_ROOT/hairfastgan_amd/csrc
HAIRFAST_HIP_LIB=$C/libhairfast_eb.so python -m pytest tests/test_gpu_parity.py -x -q -m gpu --deselect tests/test_gpu_parity.py::test_native_library_is_loaded 2>&1 | tail -2
for v in hip eb hip eb hip eb; do echo "== $v"; HAIRFAST_HIP_LIB=$C/libhairfast_$v.so python bench.py --no-cpu-baseline --no-exact-f32 --swap-triples 0 --steps 40 --warmup 5 --no-kernel-events | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06bm_early_barrier.txt
for v in hip eb; do echo "== $v"; HAIRFAST_HIP_LIB=$C/libhairfast_$v.so PROBE_TUNE=0 python tools/probes/gen_layers.py 2>&1 | grep "same\|upfu\|up2p 512"; done | tee -a gpurun_out/r06bm_early_barrier.txt
