cd $GRAFT_REPO_ROOT
C=$GRAFT_REPO_ROOT/hairfastgan_amd/csrc
PROBE_TUNE=0 python tools/probes/gen_layers.py > /dev/null 2>&1
for v in hip ab4 ab12 hip; do echo "== $v"; HAIRFAST_HIP_LIB=$C/libhairfast_$v.so PROBE_TUNE=0 python tools/probes/gen_layers.py 2>&1 | grep "same\|upfu"; done
