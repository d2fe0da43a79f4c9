cd $GRAFT_REPO_ROOT
python tools/probes/encoder_launches.py 2>&1 | grep -v "amdgpu\|Warn\|warn" | tail -44
