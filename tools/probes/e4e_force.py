"""e4e / FS-encoder forward time at batch 3 with the stride-1 3x3 tile configuration forced
(hf_debug_set_dispatch): which instantiation suits the encoder shapes?"""
import argparse, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hairfastgan_amd.encoders import Encoder4Editing, FSEncoder
from hairfastgan_amd._runtime import lib
from oracle import cases as C, ref_encoders as E
dev = torch.device("cuda:0")
L = lib()
e4e = Encoder4Editing(50, "ir_se", argparse.Namespace(stylegan_size=1024)).eval()
e4e.load_state_dict(C.params_from_shapes("e4e", E.e4e_param_shapes())); e4e = e4e.to(dev)
fs = FSEncoder(); fs.enc.load_state_dict(C.params_from_shapes("fs", E.fs_param_shapes())); fs = fs.to(dev)
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
with torch.inference_mode():
    x = torch.randn(3, 3, 256, 256, device=dev); img = torch.randn(3, 3, 1024, 1024, device=dev)
    for cfg in (0, 11, 12, 13, 14, 15, 16, 31, 32, 33, 34):
        L.hf_debug_set_dispatch(cfg, 0)
        try:
            t1 = timeit(lambda: e4e(x)); t2 = timeit(lambda: fs.test(img=img, return_latent=True))
            print(f"cfg {cfg:3d}: e4e {t1:6.2f} ms   fs {t2:6.2f} ms", flush=True)
        except Exception as ex:
            print(f"cfg {cfg:3d}: {type(ex).__name__} {str(ex)[:80]}")
    L.hf_debug_set_dispatch(0, 0)
