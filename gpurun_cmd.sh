set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -40 > gpurun_out/r02d_tests.log
HF_FORCE_DIST=1 MASTER_PORT=29711 python bench.py --workload swap256 --triples 32 --warmup 2 > gpurun_out/r02d_swap32.log 2>gpurun_out/r02d_swap32.err
python - > gpurun_out/r02d_pp_time.log 2>&1 <<'PY'
import sys, torch, time
sys.path.insert(0, '.')
import bench
from hairfastgan_amd.encoders import PostProcessModel
from oracle import ref_postprocess as PP
dev = torch.device('cuda:0')
sh = PP.post_process_param_shapes(); sh.pop('latent_avg')
pp = PostProcessModel(); pp.load_state_dict(bench.synth_state('pp', sh)); pp = pp.eval().to(dev)
a, b = torch.randn(1,3,256,256,device=dev)*0.5, torch.randn(1,3,256,256,device=dev)*0.5
from hairfastgan_amd import _runtime
for mode in ('f16x3','f32','f16'):
    _runtime.set_conv_precision(mode)
    pp(a,b); torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(5): pp(a,b)
    torch.cuda.synchronize(); t=(time.perf_counter()-t0)/5
    print(f'PostProcess {mode}: {t*1e3:.2f} ms ({774.0/t/1e3:.1f} TFLOP/s)')
PY
tail -6 gpurun_out/r02d_tests.log; cat gpurun_out/r02d_pp_time.log; head -c 600 gpurun_out/r02d_swap32.log; tail -3 gpurun_out/r02d_swap32.err
