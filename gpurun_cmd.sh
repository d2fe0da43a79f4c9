# GPU call r06y: host-side profile of one eager swap
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/probes/swap_host_profile.py > gpurun_out/r06y_swap_host_profile.txt 2>&1
grep -v amdgpu gpurun_out/r06y_swap_host_profile.txt | head -70
