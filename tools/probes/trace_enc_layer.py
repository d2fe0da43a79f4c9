"""Where a block of conv_enc_h<...,pre> spends its time (HF_ENC_TRACE build of csrc/convh_enc.hip through HAIRFAST_HIP_LIB):
python trace_enc_layer.py B cin cout H W [stride]  ->  shader clocks between the stamps of waves 0 / 7 in three tiles of the launch.
Build (from hairfastgan_amd/csrc, after build.sh):
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DHF_ENC_TRACE -c convh_enc.hip -o /tmp/enc_trace.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o libhairfast_enctrace.so $(ls *.o | grep -v convh_enc.o) /tmp/enc_trace.o
Results of round 5: profiles/r05v_trace_enc_block_before.txt, profiles/r05x_trace_enc_block.txt (lesson 24)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hairfastgan_amd import _marshal as M  # noqa: E402
from hairfastgan_amd._runtime import lib, stream  # noqa: E402

NAMES = ["entry", "index math", "LDS zero fill", "first stage landed", "K loop", "epilogue issued", "stores acknowledged"]


def main():
    B, cin, cout, H, W = (int(v) for v in sys.argv[1:6])
    stride = int(sys.argv[6]) if len(sys.argv) > 6 else 1
    dev = torch.device("cuda:0")
    L, st = lib(), stream()
    x = torch.randn(B, cin, H, W, device=dev)
    w = torch.randn(cout, cin, 3, 3, device=dev) / (cin * 9) ** 0.5
    hi, lo = M.conv_split_weights_f16(L, st, M.conv_prepare(L, st, w))
    bias = torch.randn(cout, device=dev)
    xs = M.split_activation_f16(L, st, x)
    for _ in range(3):
        M.conv2d_f16(L, st, xs, hi, lo, 3, cout, stride, bias=bias, act=M.ACT_LRELU, alpha=0.01)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    M.conv2d_f16(L, st, xs, hi, lo, 3, cout, stride, bias=bias, act=M.ACT_LRELU, alpha=0.01)
    e1.record()
    torch.cuda.synchronize()
    print(f"B={B} {cin}->{cout} @{H}x{W} s{stride}: path {L.hf_debug_last_path()}, {e0.elapsed_time(e1) * 1e3:.1f} us")
    buf = (ctypes.c_ulonglong * 48)()
    fn = L.hf_debug_read_enc_trace
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
    print("rc", fn(buf, 48))
    for sel, blk in enumerate((0, 700, 2000)):
        for wi, wv in enumerate((0, 7)):
            t = [buf[(sel * 2 + wi) * 8 + i] for i in range(7)]
            if not t[0]:
                continue
            d = [t[i + 1] - t[i] for i in range(6)]
            print(f"  block {blk:4d} wave {wv}: " + ", ".join(f"{NAMES[i + 1]} {d[i]}" for i in range(6)) + f"  | total {t[6] - t[0]} ticks")


if __name__ == "__main__":
    main()
