#!/bin/bash
# PMC passes over one conv layer (tools/probes/one_layer.py); run on the GPU box from the repo root:
#   tools/pmc_layer.sh <tag> <cin> <cout> <res> [batch] [up]
# Counters go in separate runs (slots per block are limited), each under its own timeout (a
# TA_*_sum pass once hung for the whole gpurun limit); outputs under gpurun_out/pmc_<tag>/.
tag=$1; shift
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_$tag
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY" \
           "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "FETCH_SIZE" \
           "WRITE_SIZE"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$tag/p$i -o l -- python $R/tools/probes/one_layer.py "$@" > $R/gpurun_out/pmc_$tag/p$i.log 2>&1
  echo "pass $i rc=$?"
done
