import argparse, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hairfastgan_amd.encoders import Encoder4Editing, FSEncoder
from oracle import cases as C, ref_encoders as E
dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "e4e"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 3
with torch.inference_mode():
    if which == "e4e":
        m = Encoder4Editing(50, "ir_se", argparse.Namespace(stylegan_size=1024)).eval()
        m.load_state_dict(C.params_from_shapes("e4e", E.e4e_param_shapes())); m = m.to(dev)
        x = torch.randn(B, 3, 256, 256, device=dev)
        for _ in range(3): m(x)
    else:
        m = FSEncoder(); m.enc.load_state_dict(C.params_from_shapes("fs", E.fs_param_shapes())); m = m.to(dev)
        x = torch.randn(B, 3, 1024, 1024, device=dev)
        for _ in range(3): m.test(img=x, return_latent=True)
    torch.cuda.synchronize()
