#!/bin/bash
# rocprofv3 passes over bench.py on the GPU box (run from the repo root through gpurun):
#   tools/profile_bench.sh <tag>
# 1. --kernel-trace --stats of `bench.py --steps 3 --warmup 1` (generator workload, no secondary measurements)
# 2-4. --pmc FETCH_SIZE / WRITE_SIZE / SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE in separate runs (--steps 1)
# 5. --kernel-trace --stats of the batched swap (bench.py --workload swap256 --triples 16)
# Outputs under gpurun_out/prof_<tag>_*; summarise with tools/summarize_prof.py / tools/make_pmc_traffic.py.
tag=$1
R=$GRAFT_REPO_ROOT
B="python $R/bench.py --no-cpu-baseline --no-exact-f32 --swap-triples 0"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${tag}_stats -o bench -- $B --steps 3 --warmup 1 > $R/gpurun_out/prof_${tag}_stats.log 2>&1
echo "stats rc=$?"
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  name=$(echo $set | cut -d' ' -f1); [ "$name" = "SQ_VALU_MFMA_BUSY_CYCLES" ] && name=MFMA
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/prof_${tag}_$name -o bench -- $B --steps 1 --warmup 1 --no-kernel-events > $R/gpurun_out/prof_${tag}_$name.log 2>&1
  echo "pmc $name rc=$?"
done
# (the batched swap: tools/prof_swap.sh)

