cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
C=$GRAFT_REPO_ROOT/hairfastgan_amd/csrc
for v in hip varA varB varC hip; do
  echo "== $v"; HAIRFAST_HIP_LIB=$C/libhairfast_$v.so PROBE_TUNE=0 python tools/probes/gen_layers.py 2>&1 | grep -v amdgpu
done > gpurun_out/r02p_variants.log 2>&1
cat gpurun_out/r02p_variants.log
