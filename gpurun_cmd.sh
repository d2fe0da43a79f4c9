cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/probes/time_sites.py 32 400 > gpurun_out/r05o_sites32.log 2>&1; head -4 gpurun_out/r05o_sites32.log | cut -c1-600
