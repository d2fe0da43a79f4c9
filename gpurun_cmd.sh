cd $GRAFT_REPO_ROOT
ENC_BATCH_MULT=8 python tools/bench_enc_layers.py 2>&1 | grep -v amdgpu > /dev/null
for t in 0 $((255<<16)) $((64<<16)) $((96<<16)); do echo "== tune $t"; ENC_BATCH_MULT=8 ENC_TUNE=$t python tools/bench_enc_layers.py 2>&1 | grep -v amdgpu | cut -c1-82 | grep -v "heads\|pp 1024\|pp 768"; done
