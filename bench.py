#!/usr/bin/env python3
"""bench.py - StyleGAN2 1024^2 generator forward on MI355X (BASELINE.json configs[1]).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch 8]

A "step" is one full generator forward (range 0->8) of one batch of W+ latents that are
already resident in HBM; noise is drawn fresh per layer like HairFast's callers do
(noise=None, models/stylegan2/model.py:289-291).  Weights are the closed-form synthetic
fill of oracle/synth.py (no checkpoints / network on the box); timing is weight-independent.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline      dominant kernel = the modulated-conv instantiation with the largest share of the
                step (per-launch labels come from hf_debug_last_path): ALGORITHMIC FLOPs of its
                launches / their HIP-event durations, measured during the timed steps on the launch
                stream.  peak: 157.3 TFLOP/s for the fp32-MFMA kernels; for the split-operand
                fp16 kernels (3 fp16 MFMAs per fp32-class product) the fp16 dense peak / 3.
  exact_f32     the same forward with every conv on the fp32 MFMA (HAIRFAST_CONV_PRECISION=f32),
                timed after the headline run, for comparison.
  cpu_baseline  the CPU oracle (bit-identical restatement of the reference's PyTorch CPU
                path) timed on this host: batch-1 forwards, all host cores (rank 0, N=1 only).
For N>1 (torchrun, one rank per GPU over RCCL) every rank runs the same per-GPU batch
(weak scaling) and the uint8 result images of each step are all-gathered (async, overlapped
with the next step's compute).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CU x 2.4 GHz
PEAK_F16_MFMA_TFLOPS = 2516.6  # v_mfma_f32_32x32x16_f16: 32768 FLOP / 32 cycles / SIMD, 1024 SIMDs x 2.4 GHz
DTYPES = {
    "f16x3": "f32 tensors + f32 accumulate; conv products as 3 fp16 MFMAs on (hi,lo)-split operands (fp32-class)",
    "f32": "f32 (fp32 MFMA)",
    "f16": "f32 tensors + f32 accumulate; conv operands rounded to fp16 (fp16 MFMA)",
}
GFLOP_PER_IMAGE = 148.52       # modulated-conv FLOPs of one 0->8 forward


def build_generator(dev):
    import numpy as np

    from hairfastgan_amd.stylegan2.model import Generator
    from oracle import synth  # closed-form parameter fill only (data, not compute)

    g = Generator(1024, 512, 8, channel_multiplier=2).eval()
    sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in
          synth.fill_state_dict({k: tuple(v.shape) for k, v in g.state_dict().items()}).items()}
    g.load_state_dict(sd)
    return g.to(dev), sd


def cpu_baseline(sd, budget_s=30.0):
    """Oracle forward (generator 0->8, batch 1, explicit noise) timed on this host with all
    cores and, because very wide hosts oversubscribe ATen's grouped convs, with 32 threads;
    the faster setting is reported with the thread count it used."""
    from oracle import cases as C
    from oracle import ref_stylegan2 as O

    ncpu = os.cpu_count() or 1
    lat, nz, _ = C.generator_inputs(1024, 1, 0)
    tried = {}
    t_start = time.time()
    # very wide hosts (256 hardware threads) take ~40 s per forward with all threads: only try
    # the full count up to 64 cores, otherwise 32 threads
    counts = {min(ncpu, 32)} | ({ncpu} if ncpu <= 64 else set())
    for threads in sorted(counts, reverse=True):
        torch.set_num_threads(threads)
        times = []
        with torch.inference_mode():
            O.generator_forward(sd, lat, nz)  # warm-up
            while len(times) < 3 and (not times or (time.time() - t_start) < budget_s):
                t0 = time.time()
                O.generator_forward(sd, lat, nz)
                times.append(time.time() - t0)
        times.sort()
        tried[threads] = (times[len(times) // 2], len(times))
    cores = min(tried, key=lambda k: tried[k][0])
    med, n = tried[cores]
    return {"value": round(1.0 / med, 4), "unit": "images/s", "cores": cores, "kind": "port",
            "ms_per_image": round(med * 1e3, 1),
            "sample": f"oracle (CPU restatement, bit-identical to the reference's PyTorch CPU path) generator "
                      f"0->8, batch 1, explicit noise; median of {n} timed forwards after 1 warm-up, torch "
                      f"{torch.__version__}; thread counts tried: "
                      + ", ".join(f"{k}T={v[0] * 1e3:.0f}ms" for k, v in sorted(tried.items()))}


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/pmc_traffic.json:
    FETCH_SIZE + WRITE_SIZE collected in separate rocprofv3 --pmc runs of this bench), or None."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            k = json.load(f)["kernels"].get(kernel)
        return None if k is None else round(k["fetch_bytes_per_launch"] + k["write_bytes_per_launch"])
    except (OSError, KeyError, ValueError):
        return None


def pmc_mfma_busy(kernel):
    """MFMA-busy fraction of `kernel` at the clock it actually ran at (same committed PMC passes), or None."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            return json.load(f)["kernels"].get(kernel, {}).get("mfma_busy")
    except (OSError, ValueError, KeyError):
        return None


def swap_schedule_bench(g, sd, dev, n_triples):
    """Seconds for `n_triples` replays of the per-triple hot-path schedule (after one warm-up)."""
    import numpy as np

    from hairfastgan_amd.hair_swap import HairFastHotPath, get_parser
    from oracle import ref_encoders as E
    from oracle import synth

    def fill(prefix, shapes):
        return {k: torch.from_numpy(np.ascontiguousarray(synth.fill_value(f"{prefix}.{k}", tuple(s)))) for k, s in shapes.items()}

    args = get_parser().parse_args([])
    args.device = dev
    hp = HairFastHotPath(args, {"g_ema": sd, "latent_avg": torch.zeros(512)}, fill("e4e", E.e4e_param_shapes()),
                         fill("fs", E.fs_param_shapes()))
    torch.manual_seed(3407)
    z = lambda *s: torch.randn(*s, device=dev)  # noqa: E731
    inputs = (z(3, 3, 1024, 1024) * 0.5, z(3, 3, 256, 256) * 0.5, z(2, 3, 256, 256) * 0.5, z(1, 512, 32, 32),
              z(1, 512, 64, 64), z(1, 18, 512), z(1, 18, 512), z(1, 18, 512))
    res = {}
    for mode, ug in (("eager", False), ("hipgraph", True)):
        try:
            hp.swap_schedule(*inputs, use_graphs=ug)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n_triples):
                hp.swap_schedule(*inputs, use_graphs=ug)
            torch.cuda.synchronize()
            res[mode] = time.perf_counter() - t0
        except Exception as e:  # graph capture problems must not take the headline number down
            res[mode] = None
            res[mode + "_error"] = f"{type(e).__name__}: {e}"[:200]
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=8, help="images per GPU per step")
    ap.add_argument("--precision", choices=("f16x3", "f32", "f16"), default=None,
                    help="matrix-core mode of the 3x3 convs (default: HAIRFAST_CONV_PRECISION or f16x3)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-exact-f32", action="store_true", help="skip the fp32-MFMA comparison run (profiling)")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--swap-triples", type=int, default=4,
                    help="triples per GPU for the secondary hair-swap hot-path schedule measurement (0 = skip)")
    args = ap.parse_args()

    from hairfastgan_amd import _marshal, _runtime, parallel

    if args.precision:
        _runtime.set_conv_precision(args.precision)
    precision = _runtime.conv_precision()

    rank, world, local = parallel.init_from_env()
    if world != args.gpus and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    g, sd = build_generator(dev)
    B = args.batch
    torch.manual_seed(3407 + rank)  # the reference's default seed (utils/seed.py:19)
    latent = torch.randn(B, 18, 512, device=dev)

    import torch.distributed as dist

    gather_note = None
    pending = None

    use_dist = world > 1 or os.environ.get("HF_FORCE_DIST", "0") == "1"

    def step():
        nonlocal pending, gather_note
        with torch.inference_mode():
            img, _ = g([latent], input_is_latent=True)
            if use_dist:
                try:
                    u8 = parallel.to_uint8_image(img)
                    if pending is not None:
                        pending[1].wait()
                    pending = parallel.all_gather_images(u8, async_op=True)
                except Exception as e:  # recorded in the JSON, never silent
                    gather_note = f"all_gather failed: {type(e).__name__}: {e}"
        return img

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    if pending is not None:
        pending[1].wait()
    barrier()
    prof = None if args.no_kernel_events else []
    _marshal.PROFILE = prof
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if pending is not None:
        pending[1].wait()
    barrier()
    elapsed = time.perf_counter() - t0
    _marshal.PROFILE = None

    if use_dist:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # comparison runs (outside the timed region): the same forward with every conv on the exact-fp32
    # MFMA, and BASELINE.json configs[4] (fp16 operands, batch 16)
    def alt_run(mode, batch):
        prev = _runtime.set_conv_precision(mode)
        lat = torch.randn(batch, 18, 512, device=dev)
        try:
            with torch.inference_mode():
                for _ in range(2):
                    g([lat], input_is_latent=True)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    g([lat], input_is_latent=True)
                torch.cuda.synchronize()
                return time.perf_counter() - t1
        finally:
            _runtime.set_conv_precision(prev)

    exact_f32 = f16_mode = None
    if world == 1 and not args.no_exact_f32:
        if precision != "f32":
            e32 = alt_run("f32", B)
            exact_f32 = {"value": round(B * args.steps / e32, 3), "unit": "images/s", "ms_per_step": round(e32 / args.steps * 1e3, 4),
                         "note": "same forward, HAIRFAST_CONV_PRECISION=f32 (v_mfma_f32_32x32x2_f32 only)"}
        e16 = alt_run("f16", 16)
        f16_mode = {"value": round(16 * args.steps / e16, 3), "unit": "images/s", "ms_per_step": round(e16 / args.steps * 1e3, 4),
                    "batch": 16, "note": "BASELINE.json configs[4]: fp16 conv operands (HAIRFAST_CONV_PRECISION=f16; the hand-over "
                                         "activations between convs are fp16, everything else fp32), fp32 accumulation and "
                                         "demodulation; pixel MSE vs the reference 2e-6 (tests/test_gpu_parity.py)"}

    # Secondary measurement (outside the timed region above): the hot-path call schedule of
    # one HairFast swap (BASELINE.json configs[2]/[3]), triples sharded
    # over ranks, no collective in the path.
    swap_info = None
    if args.swap_triples > 0:
        swap_res = swap_schedule_bench(g, sd, dev, args.swap_triples)
        times = {}
        for mode in ("eager", "hipgraph"):
            tsec = swap_res.get(mode)
            if tsec is not None and use_dist:
                t = torch.tensor([tsec], device=dev, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                tsec = float(t.item())
            times[mode] = tsec
        best = min((v for v in times.values() if v is not None), default=None)
        swap_info = {"metric": "hair_swap_hot_path_triples_per_sec",
                     "value": None if best is None else round(args.swap_triples * world / best, 3),
                     "unit": "triples/s",
                     "ms_per_triple": {m: (None if v is None else round(v / args.swap_triples * 1e3, 2)) for m, v in times.items()},
                     "errors": {k: v for k, v in swap_res.items() if k.endswith("_error")},
                     "triples_per_gpu": args.swap_triples,
                     "workload": "per triple: e4e B=3, FS-encoder B=3, gen 3->3 B=3, gen 0->3 B=3, gen 0->8 B=1, e4e B=2, "
                                 "gen 0->3 B=2, gen 0->8 B=1, gen 4->8 B=1, gen 5->8 B=1 (1545 GFLOP; the reference's "
                                 "discarded FS-encoder generator forward is not run); BiSeNet/SEAN/CLIP/PostProcess "
                                 "stages are out of scope and replaced by resident synthetic tensors",
                     "gflop_per_triple": 1545}

    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        value = B * args.steps * world / elapsed
        out = {
            "metric": "stylegan2_generator_fwd_1024_images_per_sec", "value": round(value, 3), "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 4),
            "ms_per_image": round(ms_step / B, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": DTYPES[precision], "data": "synthetic",
            "config": {"workload": "StyleGAN2 1024^2 generator forward (range 0->8), random W+, fresh noise per layer, "
                                   "HIP modulated-conv + upfirdn2d kernels (BASELINE.json configs[1])",
                       "batch_per_gpu": B, "global_batch": B * world, "parallelism": f"replica x{world}",
                       "weights": "synthetic closed-form (oracle/synth.py)", "conv_precision": precision},
            "algorithmic_tflops_whole_forward": round(value * GFLOP_PER_IMAGE / 1e3 / world, 2),
        }
        if gather_note:
            out["config"]["gather"] = gather_note
        elif use_dist:
            out["config"]["gather"] = "async RCCL all_gather_into_tensor of uint8 images, one per step"
        if prof:
            agg = {}
            for label, flops, e0, e1 in prof:
                a = agg.setdefault(label, [0.0, 0.0, 0])
                a[0] += flops
                a[1] += e0.elapsed_time(e1) * 1e-3
                a[2] += 1
            fams = {k: {"tflops": round(v[0] / v[1] / 1e12, 2), "avg_launch_ms": round(v[1] / v[2] * 1e3, 4),
                        "launches": v[2], "share_of_step": round(v[1] / elapsed, 4)} for k, v in agg.items()}
            dom = max(agg, key=lambda k: agg[k][1])
            d = agg[dom]
            ach = d[0] / d[1] / 1e12
            if dom.startswith("conv_mfma_h"):
                terms = 3 if precision == "f16x3" else 1
                peak = PEAK_F16_MFMA_TFLOPS / terms
                peak_note = (f"fp16 dense MFMA peak {PEAK_F16_MFMA_TFLOPS} TFLOP/s / {terms} MFMA per product; "
                             f"MFMA FLOPs issued = {terms} x algorithmic = {round(ach * terms, 1)} TFLOP/s")
            else:
                peak, peak_note = PEAK_FP32_MFMA_TFLOPS, "fp32 MFMA peak"
            out["roofline"] = {"bound": "mfma", "kernel": dom, "achieved": round(ach, 2),
                               "peak": round(peak, 1), "unit": "TFLOP/s",
                               "frac": round(ach / peak, 4), "traffic": pmc_traffic(dom),
                               "avg_launch_ms": round(d[1] / d[2] * 1e3, 4), "launches": d[2],
                               "flops_per_launch_avg": d[0] / d[2], "peak_note": peak_note,
                               "mfma_busy_pmc": pmc_mfma_busy(dom)}
            out["kernels"] = fams
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sd)
        if exact_f32 is not None:
            out["exact_f32"] = exact_f32
        if f16_mode is not None:
            out["f16_mode"] = f16_mode
        if swap_info is not None:
            out["swap_schedule"] = swap_info
        print(json.dumps(out), flush=True)

    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
