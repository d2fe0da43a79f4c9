cd $GRAFT_REPO_ROOT
PROBE_TUNE=0 python tools/probes/gen_layers.py > /dev/null 2>&1
PROBE_TUNE=0,8,0,8 python tools/probes/gen_layers.py 2>&1 | grep -v amdgpu
