#!/bin/bash
# rocprofv3 passes over the batched swap at the pass size the bench line is timed on (run on the GPU box from the repo root):
#   tools/prof_swap.sh <tag> [stats|pmc|det ...]     (default: stats)
# stats: --kernel-trace --stats;  det / nodet: the same with HAIRFAST_DETERMINISTIC=1 / 0;  pmc: FETCH_SIZE / WRITE_SIZE / MFMA-busy in
# separate runs (counters never together with other trace domains).  bench.py brackets its timed region with
# hf_profile_marker_kernel launches; tools/summarize_prof.py --between / tools/make_pmc_traffic.py --between cut to it.
tag=$1; shift
what=${*:-stats}
R=$GRAFT_REPO_ROOT
B="python $R/bench.py --workload swap256 --triples ${PROF_TRIPLES:-32} --swap-batch ${PROF_SWAP_BATCH:-32} --warmup 1 --no-kernel-events --no-verify"
cd /tmp && export TMPDIR=/tmp
for w in $what; do
  case $w in
    stats)
      timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${tag}_swap_stats -o bench -- $B > $R/gpurun_out/prof_${tag}_swap_stats.log 2>&1
      echo "swap stats rc=$?"; tail -c 300 $R/gpurun_out/prof_${tag}_swap_stats.log | head -c 200; echo ;;
    det)
      HAIRFAST_DETERMINISTIC=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${tag}_swapdet_stats -o bench -- $B > $R/gpurun_out/prof_${tag}_swapdet_stats.log 2>&1
      echo "swap det stats rc=$?" ;;
    nodet)  # plans from the whole launch (the opt-out of the batch-invariant default)
      HAIRFAST_DETERMINISTIC=0 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${tag}_swapnodet_stats -o bench -- $B > $R/gpurun_out/prof_${tag}_swapnodet_stats.log 2>&1
      echo "swap nodet stats rc=$?" ;;
    pmc)
      for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
        name=$(echo $set | cut -d' ' -f1); [ "$name" = "SQ_VALU_MFMA_BUSY_CYCLES" ] && name=MFMA
        timeout 900 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/prof_${tag}_swap_$name -o bench -- $B > $R/gpurun_out/prof_${tag}_swap_$name.log 2>&1
        echo "swap pmc $name rc=$?"
      done ;;
  esac
done
