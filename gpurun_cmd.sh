# GPU call r05e: MFMA issue-rate micro-benchmark (tools/probes/mfma_rate.hip)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
./tools/probes/bin/mfma_rate > gpurun_out/r05e_mfma_rate.log 2>&1; cat gpurun_out/r05e_mfma_rate.log
