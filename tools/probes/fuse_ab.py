"""The three one-kernel upsampling StyledConv layers (pre-split input, split output, batch 8) timed under every library named on
the command line (A/B builds of tools/build_one.sh / build_variant.sh): python fuse_ab.py hip var1 var2:32 ...  (name[:tuning bits] -> libhairfast_<name>.so under hf_debug_set_tuning).
Each library runs in its own process (the library is bound at import)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))

CHILD = r'''
import os, sys, torch
sys.path.insert(0, %r)
from hairfastgan_amd import _marshal as M
from hairfastgan_amd._runtime import lib, stream
from oracle import ref_stylegan2 as O
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
dev = torch.device("cuda:0"); L, st = lib(), stream()
L.hf_debug_set_tuning(int(os.environ.get("PROBE_TUNE", "0")))
k4 = O.blur_kernel_1d_to_2d(gain=4.0).to(dev); fac = M.blur_factors(k4)
B = int(os.environ.get("PROBE_BATCH", "8"))
out = []
for cin, cout, h in [(64, 32, 512), (128, 64, 256), (256, 128, 128)]:
    torch.manual_seed(0)
    x = torch.randn(B, cin, h, h, device=dev)
    wgt = torch.randn(1, cout, cin, 3, 3, device=dev)
    s, d = torch.rand(B, cin, device=dev) + 0.5, torch.rand(B, cout, device=dev) + 0.5
    s2 = torch.rand(B, cout, device=dev) + 0.5
    nz, nw, bias = torch.randn(B, 1, 2 * h, 2 * h, device=dev), torch.tensor([0.3], device=dev), torch.randn(cout, device=dev)
    wt, _ = M.prepare_weights(L, st, wgt)
    hi, lo = M.split_weights_f16(L, st, wt)
    xs = M.SplitActivation(*M.split_activation_reference(x, s), None)
    del x
    out.append(timeit(lambda: M.modconv3x3_up_fused(L, st, xs, hi, lo, None, d, fac, nz, nw, bias, split_for=s2)))
print("RESULT " + " ".join(f"{t:8.1f}" for t in out) + f"   avg {sum(out) / 3:8.1f}")
''' % ROOT

for rep in range(int(os.environ.get("PROBE_REPS", "2"))):
    for name in sys.argv[1:]:
        env = dict(os.environ)
        libname, _, tune = name.partition(":")  # "hip:32" = libhairfast_hip.so under hf_debug_set_tuning(32)
        env["PROBE_TUNE"] = tune or "0"
        env["HAIRFAST_HIP_LIB"] = os.path.join(ROOT, "hairfastgan_amd", "csrc", f"libhairfast_{libname}.so")
        r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
        print(f"{name:16s} 64->32@512 / 128->64@256 / 256->128@128 us: {line[0][7:] if line else 'FAILED ' + r.stderr[-300:]}", flush=True)
