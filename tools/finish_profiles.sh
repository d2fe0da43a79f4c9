#!/bin/bash
# After tools/profile_all.sh <tag> came back (gpurun_out/prof_<tag>_*): the committed summaries under profiles/.
tag=$1
cd "$(dirname "$0")/.."
G=gpurun_out
PROFILE_CMD="python bench.py --no-cpu-baseline --no-exact-f32 --swap-triples 0 --steps 8 --warmup 2; counters: --steps 2 --warmup 1 --no-kernel-events" python tools/summarize_prof.py --between $tag $G/prof_${tag}_stats $G/prof_${tag}_FETCH_SIZE $G/prof_${tag}_WRITE_SIZE $G/prof_${tag}_MFMA > profiles/${tag}_rocprof_summary.md
cp $G/prof_${tag}_stats/bench_kernel_stats.csv profiles/${tag}_kernel_stats.csv 2>/dev/null || cp $(find $G/prof_${tag}_stats -name "*kernel_stats.csv" | head -1) profiles/${tag}_kernel_stats.csv
PROFILE_TAG=$tag PROFILE_CMD="python bench.py --no-cpu-baseline --no-exact-f32 --swap-triples 0 --steps 2 --warmup 1 --no-kernel-events" \
  python tools/make_pmc_traffic.py --between $G/prof_${tag}_FETCH_SIZE $G/prof_${tag}_WRITE_SIZE $G/prof_${tag}_MFMA $G/prof_${tag}_stats > profiles/pmc_traffic.json
PROFILE_CMD="python bench.py --workload swap256 --triples 32 --swap-batch 32 --warmup 1 --no-kernel-events --no-verify" python tools/summarize_prof.py --between ${tag}_swap $G/prof_${tag}_swap_stats $G/prof_${tag}_swap_FETCH_SIZE $G/prof_${tag}_swap_WRITE_SIZE $G/prof_${tag}_swap_MFMA > profiles/${tag}_swap_kernel_stats.md
PROFILE_TAG=$tag PROFILE_SWAP_BATCH=32 PROFILE_CMD="python bench.py --workload swap256 --triples 32 --swap-batch 32 --warmup 1 --no-kernel-events --no-verify" \
  python tools/make_pmc_traffic.py --between $G/prof_${tag}_swap_FETCH_SIZE $G/prof_${tag}_swap_WRITE_SIZE $G/prof_${tag}_swap_MFMA $G/prof_${tag}_swap_stats > profiles/pmc_traffic_swap.json
head -30 profiles/${tag}_rocprof_summary.md
