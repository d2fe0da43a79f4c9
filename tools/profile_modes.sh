#!/bin/bash
# rocprofv3 stats + PMC passes of the generator workload in the two comparison modes of the bench line (run on the GPU box from the repo
# root through gpurun): plain fp16 operands at batch 16 (BASELINE.json configs[4]) and exact fp32 products at batch 8.
#   tools/profile_modes.sh <tag>   ->  gpurun_out/prof_<tag>_{f16,f32}_{stats,FETCH_SIZE,WRITE_SIZE,MFMA}
tag=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for mode in f16 f32; do
  batch=8; [ "$mode" = "f16" ] && batch=16
  B="python $R/bench.py --no-cpu-baseline --no-exact-f32 --swap-triples 0 --precision $mode --batch $batch --priming 2"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${tag}_${mode}_stats -o bench -- $B --steps 8 --warmup 2 --no-kernel-events > $R/gpurun_out/prof_${tag}_${mode}_stats.log 2>&1
  echo "$mode stats rc=$?"
  for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    name=$(echo $set | cut -d' ' -f1); [ "$name" = "SQ_VALU_MFMA_BUSY_CYCLES" ] && name=MFMA
    timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/prof_${tag}_${mode}_$name -o bench -- $B --steps 2 --warmup 1 --no-kernel-events > $R/gpurun_out/prof_${tag}_${mode}_$name.log 2>&1
    echo "$mode pmc $name rc=$?"
  done
done
