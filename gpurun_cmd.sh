# GPU call r06p: clock / power telemetry of OUR GPU (PCI address of HIP device 0)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r06p_clock_power.txt
{
echo "# sysfs telemetry (tools/probes/clock_power.py) of HIP device 0 - one MI355X box"
python -c "import torch; p = torch.cuda.get_device_properties(0); print(p)"
timeout 60 rocm-smi --showbus 2>&1 | grep -i "GPU\[" | head
echo "== MFMA-only probe, 200000 iterations (0.4-0.8 s per mode), non-zero operands"
python tools/probes/clock_power.py --interval 0.02 --label mfma200k -- tools/probes/bin/mfma_rate 200000 0
echo "== MFMA-only probe, 200000 iterations, ZERO operands"
python tools/probes/clock_power.py --interval 0.02 --label mfma200k-zero -- tools/probes/bin/mfma_rate 200000 1
echo "== bench.py generator workload, 300 timed steps, no per-kernel events"
python tools/probes/clock_power.py --interval 0.02 --label bench-generator -- python bench.py --no-cpu-baseline --no-exact-f32 --swap-triples 0 --steps 300 --warmup 5 --no-kernel-events
echo "== bench.py swap256 workload, 64 triples at 32 per pass"
python tools/probes/clock_power.py --interval 0.02 --label bench-swap -- python bench.py --workload swap256 --triples 64 --swap-batch 32 --warmup 1 --no-kernel-events --no-cpu-baseline
} > $O 2>&1
grep -v "^{\|MFMA only\|ds_read" $O | cut -c1-250
