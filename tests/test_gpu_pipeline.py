"""GPU parity test (-m gpu) of ONE COMPLETE HAIR SWAP against the reference's own stage classes.

tests/golden/pipeline.npz was produced by oracle/make_pipeline_golden.py: the reference's `Embedding`, `Alignment` and
`Blending` objects (models/Embedding.py:56-117, models/Alignment.py:43-181, models/Blending.py:36-82) around the reference's
own sub-networks, run on CPU in the order of hair_swap.py:38-61, with every source of randomness replaced by a formula.
Here `HairFast.swap` runs the same swap on the MI355X - every network native, no stand-in stage - with the same formulas,
and every intermediate the golden file holds is compared: W / S / F of the three embedded images and their parsing masks,
the rotated latents, the masks of the rotated images and of the shape adaptor, SEAN's two renderings, latent_F_align,
S_blend, S_final / F_final, and the final image.  Mask indices must be EQUAL (north_star: bit-exact segmentation-mask
indices) - continuous quantities are compared at the fp32 tolerance only where all upstream masks agree."""
import functools
import os

import numpy as np
import pytest
import torch

from oracle import cases as C
from oracle import ref_encoders as E
from oracle import ref_postprocess as PP
from oracle import ref_stylegan2 as O

pytestmark = pytest.mark.gpu


DEBUG = os.environ.get("HF_PIPE_DEBUG", "0") == "1"  # report every comparison instead of stopping at the first failure


def _close(got, ref, what, tol=1e-4, rms_tol=3e-5):
    got, ref = torch.as_tensor(got).detach().cpu().double(), torch.as_tensor(ref).double()
    assert got.shape == ref.shape, (what, tuple(got.shape), tuple(ref.shape))
    err = float((got - ref).abs().max())
    scale = max(1.0, float(ref.abs().max()))
    rms = float((got - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt().clamp_min(1e-12))
    if DEBUG:
        print(f"  {what}: max-abs {err:.3e} (scale {scale:.2f}), rel-rms {rms:.3e}" + ("   <-- FAIL" if not (err <= tol * scale and rms <= rms_tol) else ""))
        return err / scale
    assert err <= tol * scale and rms <= rms_tol, (what, err, scale, rms)
    return err / scale


def _samples(x, n):
    f = x.detach().reshape(-1)
    step = max(1, f.numel() // n)
    return f[::step][:n]


def test_swap_vs_reference_stage_classes(golden):
    import hairfastgan_amd.hair_swap as HS
    from hairfastgan_amd.hair_swap import HairFast, get_parser

    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU")
    G = golden("pipeline.npz")
    dev = torch.device("cuda:0")
    args = get_parser().parse_args([])
    args.device = dev
    pp_shapes = PP.post_process_param_shapes()
    lat_shape = pp_shapes.pop("latent_avg")
    _, e4e_latent_avg = C.e4e_inputs(2)
    _, dlat = C.fs_inputs(2)
    hf = HairFast(args, generator_state={"g_ema": C.generator_params(O.generator_param_shapes(1024, 512, 8, 2)), "latent_avg": torch.zeros(512)},
                  e4e_state=C.params_from_shapes("e4e", E.e4e_param_shapes()), e4e_latent_avg=e4e_latent_avg,
                  fs_state=C.params_from_shapes("fs", E.fs_param_shapes()), fs_dlatent_avg=dlat,
                  pp_state=C.params_from_shapes("pp", pp_shapes),
                  pp_latent_avg=C.params_from_shapes("pp", {"latent_avg": lat_shape})["latent_avg"] * 0.1,
                  bisenet_state=C.pipeline_bisenet_params(), rotate_state=C.params_from_shapes("rotate", PP.rotate_param_shapes()),
                  blend_state=C.params_from_shapes("clipblend", PP.clip_blending_param_shapes()), clip_state=C.clip_params(),
                  shape_state=C.shape_adaptor_params(), sean_state=C.sean_params(), sean_mean_codes=C.sean_mean_codes())
    # --- the formulas that replace randomness on both sides ---
    gen_fwd = hf.net.generator.forward
    calls = []

    def recorded_forward(styles, **kw):
        out = gen_fwd(styles, randomize_noise=False, **kw)
        calls.append({"sig": (styles[0].shape[0], kw.get("start_layer", 0), kw.get("end_layer", 8)), "latent": styles[0],
                      "layer_in": kw.get("layer_in"), "out": out[0]})
        return out

    hf.net.generator.forward = recorded_forward
    hf.stages.sean_model.netG.noise_source = lambda d, sizes: [
        torch.cat([C.pipeline_sean_noise(dd * len(sizes) + i, r) for dd in range(d)]).to(dev) for i, r in enumerate(sizes)]
    rec = {"parses": [], "targets": [], "sean": [], "embed": None, "align": []}
    seg = HS.get_segmentation
    HS.get_segmentation = lambda net, x, **kw: (rec["parses"].append(seg(net, x, **kw)) or rec["parses"][-1])
    adaptor = hf.stages.shape_adaptor
    hf.stages.shape_adaptor = lambda a, b: (rec["targets"].append(adaptor(a, b)) or rec["targets"][-1])
    sean = hf.stages.sean_inpaint_pairs
    hf.stages.sean_inpaint_pairs = lambda *a: (rec["sean"].append(sean(*a)) or rec["sean"][-1])
    emb = hf.embed.embedding_images
    hf.embed.embedding_images = lambda *a, **k: (rec.__setitem__("embed", emb(*a, **k)) or rec["embed"])
    alb = hf.align.align_images_batch
    hf.align.align_images_batch = lambda *a, **k: (rec["align"].append(alb(*a, **k)) or rec["align"][-1])
    try:
        # float images divided on the CPU, as the reference's equal_replacer / ImagesDataset do (a GPU `x / 255` is x * (1/255))
        face, shape, color = (im.float().div(255).to(dev) for im in C.pipeline_images())
        final = hf.swap(face, shape, color)
    finally:
        HS.get_segmentation = seg
    torch.cuda.synchronize()
    report = {}
    # ---- Embedding stage (Embedding.py:64-110) ----
    flips = {}
    for n in ("face", "shape", "color"):
        e = rec["embed"][n]
        report[f"W_{n}"] = _close(e["W"][0], G[f"W_{n}"], f"W {n}")
        report[f"S_{n}"] = _close(e["S"][0], G[f"S_{n}"], f"S {n}")
        flips[f"mask_{n}"] = int((e["mask"][0, 0].cpu() != torch.from_numpy(G[f"mask_{n}"].astype(np.int64))).sum())
        _close(_samples(e["image_256"], 512), G[f"image_256_{n}_samples"], f"image_256 {n}", tol=2e-6)
    assert DEBUG or all(v == 0 for v in flips.values()), flips          # 3 x 65 536 indices: every one equal
    for n in ("face", "shape", "color"):                        # F includes the hair-mask mixing (:84-91)
        report[f"F_{n}"] = _close(rec["embed"][n]["F"][0, ::16], G[f"F_{n}_chan16"], f"F {n}")
    # ---- generator calls: reference order fs33, w03, rot_shape, sean03, rot_color, blend48, final58; here the two rotations are one call
    sig = [c["sig"] for c in calls]
    assert sig == [(3, 3, 3), (3, 0, 3), (2, 0, 8), (2, 0, 3), (1, 4, 8), (1, 5, 8)], sig
    by = {"fs33": (calls[0], None), "w03": (calls[1], None), "rot_shape": (calls[2], 0), "rot_color": (calls[2], 1),
          "sean03": (calls[3], None), "blend48": (calls[4], None), "final58": (calls[5], None)}
    # ---- Alignment: rotate stage + shape adaptor (Alignment.py:58-77) ----
    for nm in ("rot_shape", "rot_color"):
        c, row = by[nm]
        _close(c["latent"][row:row + 1], G[f"call_{nm}_latent"], f"{nm} latent")
        report[nm] = _close(_samples(c["out"][row:row + 1], 1024), G[f"call_{nm}_out_samples"], f"{nm} image")
    assert len(rec["parses"]) == 2 and len(rec["targets"]) == 1
    rot_masks, targets = rec["parses"][1], rec["targets"][0]
    for row, nm in enumerate(("shape", "color")):
        flips[f"rot_mask_{nm}"] = int((rot_masks[row, 0].cpu() != torch.from_numpy(G[f"rot_mask_{nm}"].astype(np.int64))).sum())
        flips[f"target_mask_{nm}"] = int((targets[row, 0].cpu() != torch.from_numpy(G[f"target_mask_{nm}"].astype(np.int64))).sum())
    print("mask index differences vs the reference:", flips)
    # BiSeNet's five masks (3 x 512^2 inputs, 2 x 1024^2 generated images): every index equal.  The shape adaptor's label maps
    # are an argmax over the mask generator's 19 class scores: observed 1 of 65 536 indices different per map - at a pixel where
    # the REFERENCE's own top-1 / top-2 scores are an exact tie (margin 0.0; profiles/r06r_target_mask_flips.txt: the same one
    # pixel per map with the adaptor's decoders, or the whole swap, on exact fp32 products - only the reference's own summation
    # order would reproduce its tie-break).  Allowed: <= 2 per map, only at margins <= 1e-4, and never on the hair / non-hair
    # decision (HM_X, the only thing the rest of the swap derives from these maps, must be bit-equal).
    assert DEBUG or all(v == 0 for k, v in flips.items() if not k.startswith("target_")), flips
    for row, nm in enumerate(("shape", "color")):
        diff = targets[row, 0].cpu() != torch.from_numpy(G[f"target_mask_{nm}"].astype(np.int64))
        if bool(diff.any()):
            margin = torch.from_numpy(G[f"target_margin_{nm}"].astype(np.float32))[diff]
            print(f"  target mask {nm}: {int(diff.sum())} index(es) differ, reference margin(s) {[round(float(m_), 5) for m_ in margin]}")
            assert DEBUG or (int(diff.sum()) <= 2 and float(margin.max()) <= 1e-4), (nm, int(diff.sum()), float(margin.max()))
    al = rec["align"][0][0]
    hm = np.packbits((al["HM_X"][0, 0] > 0.5).cpu().numpy().astype(np.uint8))
    assert np.array_equal(hm, G["HM_X_shape"])
    hm_c = (targets[1, 0] == 13).cpu().numpy().astype(np.uint8)
    assert np.array_equal(np.packbits(hm_c), G["HM_X_color"])
    # ---- SEAN (Alignment.py:123-131) and the F-space alignment (:133-157) ----
    for d in range(2):
        img = rec["sean"][0][d]
        report[f"sean{d}"] = _close(img[:, 96:160, 96:160], G[f"sean{d}_crop"], f"sean {d} crop")
        _close(_samples(img, 2048), G[f"sean{d}_samples"], f"sean {d} samples")
    _close(calls[3]["latent"], G["call_sean03_latent"], "e4e of the SEAN renderings")
    report["latent_F_align"] = _close(al["latent_F_align"][0, ::16], G["latent_F_align_chan16"], "latent_F_align")
    # ---- Blending (Blending.py:36-82) ----
    c = by["blend48"][0]
    report["S_blend"] = _close(c["latent"], G["call_blend48_latent"], "S_blend")
    _close(c["layer_in"][:, ::16], G["call_blend48_layer_in_chan16"], "blend48 layer_in")
    _close(_samples(c["out"], 1024), G["call_blend48_out_samples"], "I_blend")
    c = by["final58"][0]
    report["S_final"] = _close(c["latent"], G["call_final58_latent"], "S_final")
    report["F_final"] = _close(c["layer_in"][:, ::16], G["call_final58_layer_in_chan16"], "F_final")
    # ---- the final image ----
    assert final.shape == (3, 1024, 1024)
    report["final"] = _close(_samples(final, 4096), G["final_samples"], "final samples", tol=1e-4, rms_tol=1e-4)
    c0 = 512 - 32
    _close(final[:, c0:c0 + 64, c0:c0 + 64], G["final_crop"], "final crop", rms_tol=1e-4)
    for nm, (sy, sx) in C.edge_crops(1024).items():
        _close(final[:, sy, sx], G[f"final_edges_{nm}"], f"final {nm}", rms_tol=1e-4)
    st = G["final_stats"]
    assert abs(float(final.mean()) - st[0]) < 1e-5 and abs(float(final.std()) - st[1]) < 1e-5
    print("swap vs reference stage classes, max-abs / scale:", {k: f"{v:.1e}" for k, v in report.items()})
