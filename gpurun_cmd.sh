# GPU call r06bj: rocprofv3 stats + PMC of the plain-fp16 (batch 16) and exact-fp32 (batch 8) generator runs
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/profile_modes.sh r06 2>&1 | grep "rc="
