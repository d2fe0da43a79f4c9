"""Label-map flips of the CtrlHair shape adaptor against the reference golden (tests/golden/shape_adaptor.npz) by how many of the
mask decoders' trailing convolutions run with exact fp32 products (shape_adaptor.EXACT_TAIL) and in which process-wide mode:
VERDICT r05 item 7 - does exact arithmetic in the adaptor's last layers remove the 1-in-65536 index difference?"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hairfastgan_amd import _runtime, shape_adaptor as SA  # noqa: E402
from oracle import cases as C  # noqa: E402

dev = torch.device("cuda:0")
G = np.load(os.path.join(ROOT, "tests", "golden", "shape_adaptor.npz"))
gen = SA.MaskGenerator().eval()
gen.load_state_dict(C.shape_adaptor_params())
gen.to(dev)
m1, m2 = (m.to(dev) for m in C.shape_masks())
for mode in ("f16x3", "f32"):
    for tail in (0, 1, 2, 3, 8):
        if mode == "f32" and tail:
            continue
        SA.EXACT_TAIL = tail
        with _runtime.conv_precision_scope(mode):
            out = SA.adapt_shape(gen, m1, m2)
            flips, worst = [], 0.0
            for b in range(2):
                ref = torch.from_numpy(G["labels"][b].astype("int64")).to(dev)
                margin = torch.from_numpy(G[f"margin_{b}"].astype("float32")).to(dev)
                f = out[b, 0] != ref
                flips.append(int(f.sum()))
                if flips[-1]:
                    worst = max(worst, float(margin[f].max()))
                    ys, xs = torch.nonzero(f, as_tuple=True)
                    where = [(int(y), int(x), int(out[b, 0, y, x]), int(ref[y, x]), float(margin[y, x])) for y, x in zip(ys[:4], xs[:4])]
                else:
                    where = []
                print(f"mode {mode:6s} exact tail {tail}: map {b}: {flips[-1]} of 65536 indices differ {where}")
        print(f"mode {mode:6s} exact tail {tail}: flips {flips}, largest reference top-2 margin at a flip {worst:.3e}", flush=True)
