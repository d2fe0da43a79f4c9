set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
BARGS="--no-cpu-baseline --no-exact-f32 --swap-triples 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r02_stats -o bench -- python $R/bench.py --steps 3 --warmup 1 $BARGS > $R/gpurun_out/prof_r02_stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/prof_r02_$c -o bench -- python $R/bench.py --steps 1 --warmup 1 $BARGS > $R/gpurun_out/prof_r02_$c.log 2>&1
done
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/prof_r02_MFMA -o bench -- python $R/bench.py --steps 1 --warmup 1 $BARGS > $R/gpurun_out/prof_r02_MFMA.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r02_swap -o bench -- python $R/bench.py --workload swap256 --triples 6 --warmup 1 --no-kernel-events > $R/gpurun_out/prof_r02_swap.log 2>&1
cd $R
find gpurun_out/prof_r02_* -name "*.csv" | head -30
ls -la gpurun_out/prof_r02_stats/ gpurun_out/prof_r02_stats/* | head -20
