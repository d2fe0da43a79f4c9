set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/bench_enc_layers.py > gpurun_out/r02c_enc_layers.log 2>&1
python -m pytest tests/test_gpu_encoders.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r02c_enc_tests.log
python -m pytest tests/test_gpu_parity.py tests/test_gpu_schedule.py -m gpu -q -k "scaled_activations or forced_rccl" 2>&1 | tail -80 > gpurun_out/r02c_fail_tests.log
HAIRFAST_ENC_PRESPLIT=heads python tools/bench_encoders.py f16x3 > gpurun_out/r02c_encoders_heads.log 2>&1
HAIRFAST_ENC_PRESPLIT=all python tools/bench_encoders.py f16x3 > gpurun_out/r02c_encoders_all.log 2>&1
HAIRFAST_ENC_PRESPLIT=none python tools/bench_encoders.py f16x3 > gpurun_out/r02c_encoders_none.log 2>&1
cat gpurun_out/r02c_enc_layers.log; tail -3 gpurun_out/r02c_enc_tests.log; grep "B=3" gpurun_out/r02c_encoders_*.log
