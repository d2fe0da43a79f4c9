#!/bin/bash
# Every rocprofv3 pass behind profiles/<tag>_* (run on the GPU box from the repo root through gpurun): tools/profile_all.sh <tag>
#   generator workload: --kernel-trace --stats of `bench.py --steps 8 --warmup 2` (per-kernel events on every 4th step, as the driver's
#   run), then FETCH_SIZE / WRITE_SIZE / MFMA-busy counters in three separate passes (--steps 2, no events);
#   batched swap: tools/prof_swap.sh <tag> stats pmc.
# bench.py brackets both timed regions with hf_profile_marker_kernel launches; summarise with
#   tools/summarize_prof.py --between / tools/make_pmc_traffic.py --between (tools/finish_profiles.sh <tag>).
tag=$1
R=$GRAFT_REPO_ROOT
B="python $R/bench.py --no-cpu-baseline --no-exact-f32 --swap-triples 0"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${tag}_stats -o bench -- $B --steps 8 --warmup 2 > $R/gpurun_out/prof_${tag}_stats.log 2>&1
echo "stats rc=$?"
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  name=$(echo $set | cut -d' ' -f1); [ "$name" = "SQ_VALU_MFMA_BUSY_CYCLES" ] && name=MFMA
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/prof_${tag}_$name -o bench -- $B --steps 2 --warmup 1 --no-kernel-events > $R/gpurun_out/prof_${tag}_$name.log 2>&1
  echo "pmc $name rc=$?"
done
cd $R
bash tools/prof_swap.sh $tag stats pmc
