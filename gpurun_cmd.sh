# GPU call r06aj: row pipeline epilogue with grouped constant loads: hip vs base
cd $GRAFT_REPO_ROOT
C=$GRAFT_REPO_ROOT/hairfastgan_amd/csrc
for v in base hip base hip; do echo "== $v"; HAIRFAST_HIP_LIB=$C/libhairfast_$v.so PROBE_TUNE=0 python tools/probes/gen_layers.py 2>&1 | grep "same  32"; done
