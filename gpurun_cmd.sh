# GPU call r06r: the pipeline golden's target-mask flips with the shape adaptor's decoders in exact fp32 (tail 2 / all 8) and with the whole swap in f32
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r06r_target_mask_flips.txt
: > $O
for cfg in "default" "HAIRFAST_SHAPE_EXACT_TAIL=2" "HAIRFAST_SHAPE_EXACT_TAIL=8" "HAIRFAST_CONV_PRECISION=f32"; do
  echo "== $cfg" >> $O
  if [ "$cfg" = default ]; then e=""; else e="$cfg"; fi
  env $e python -m pytest tests/test_gpu_pipeline.py -x -q -m gpu -s -k "stage_classes" 2>&1 | grep -i "mask index\|target mask\|passed\|failed\|Error" >> $O
done
cat $O
