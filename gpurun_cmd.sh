cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/probes/time_sites.py gen 60 > gpurun_out/r05l_sites_gen.log 2>&1; sed -n 2,70p gpurun_out/r05l_sites_gen.log | cut -c1-210
python -m pytest tests/test_gpu_parity.py -m gpu -q -k "block_order" 2>&1 | tail -2
