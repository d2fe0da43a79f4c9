R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r3h_swapb -o bench -- python $R/bench.py --workload swap256 --triples 16 --warmup 1 --no-kernel-events > $R/gpurun_out/prof_r3h_swapb.log 2>&1
echo "batched rc=$?"; tail -c 400 $R/gpurun_out/prof_r3h_swapb.log | head -c 300; echo
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r3h_swap1 -o bench -- python $R/bench.py --workload swap256 --triples 6 --swap-batch 1 --warmup 1 --no-kernel-events > $R/gpurun_out/prof_r3h_swap1.log 2>&1
echo "single rc=$?"
ls $R/gpurun_out/prof_r3h_swapb/ | head
