import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from hairfastgan_amd import _marshal as M
from hairfastgan_amd._runtime import lib, stream
def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
dev = torch.device("cuda:0"); L, st = lib(), stream()
for B in (24, 3):
  for (cin, cout, H, G) in [(64,64,256,1),(128,128,128,1),(256,256,64,1),(512,512,32,1),(512,512,64,11)]:
    x = torch.randn(B, cin, H, H, device=dev)
    w = torch.randn(G, cout, cin, 3, 3, device=dev) / (cin * 9) ** 0.5
    wt = torch.stack([M.conv_prepare(L, st, w[g]) for g in range(G)]).contiguous()
    if G == 1: wt = wt[0]
    hi, lo = M.conv_split_weights_f16(L, st, wt)
    bias = torch.randn(G, cout, device=dev) if G > 1 else torch.randn(cout, device=dev)
    kw = dict(bias=bias, act=M.ACT_LRELU, alpha=0.01, groups=G)
    gf = 2.0 * cin * cout * 9 * (H // 2) ** 2 * B * G / 1e9
    xs = M.split_activation_f16(L, st, x)
    out = []
    ref = None
    for tune in (0, 8):
        L.hf_debug_set_tuning(tune)
        y = M.conv2d_f16(L, st, xs, hi, lo, 3, cout, 2, **kw)
        if ref is None: ref = y
        err = float((y - ref).abs().max())
        t = timeit(lambda: M.conv2d_f16(L, st, xs, hi, lo, 3, cout, 2, **kw))
        t2 = timeit(lambda: M.conv2d_f16(L, st, x, hi, lo, 3, cout, 2, **kw))
        out.append(f"tune {tune}: pre {t:7.1f} us {gf / t * 1e3:6.1f} TF/s | reg {t2:7.1f} us (diff {err:.1e})")
    L.hf_debug_set_tuning(0)
    print(f"B={B} {cin}->{cout} @{H} s2 x{G}: " + " || ".join(out), flush=True)
