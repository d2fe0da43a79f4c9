mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r01k_stats -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --swap-triples 0 > $R/gpurun_out/prof_r01k_stats.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_r01k_fetch -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-events --swap-triples 0 > $R/gpurun_out/prof_r01k_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_r01k_write -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-events --swap-triples 0 > $R/gpurun_out/prof_r01k_write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES --kernel-trace --output-format csv -d $R/gpurun_out/prof_r01k_mfma -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-events --swap-triples 0 > $R/gpurun_out/prof_r01k_mfma.log 2>&1
cd $R; tail -2 gpurun_out/prof_r01k_stats.log | cut -c1-200
