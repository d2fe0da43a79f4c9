#!/usr/bin/env python3
"""bench.py - MI355X measurements of the HairFastGAN hot path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch 8] [--workload generator|swap256]

--workload generator (default; BASELINE.json configs[1], the configuration the headline metric
  is quoted on): a "step" is one full StyleGAN2 1024^2 generator forward (range 0->8) of one batch
  of W+ latents that are already resident in HBM; noise is drawn fresh per layer like HairFast's
  callers do (noise=None, models/stylegan2/model.py:289-291).  value = images/s.
--workload swap256 (BASELINE.json configs[3]): `--triples` (default 256) synthetic 1024^2 triples,
  block-partitioned over the ranks; every triple goes host uint8 -> H2D -> HairFast.swap (the complete
  call schedule of hair_swap.py:38-61; every network runs natively, incl. SEAN and the CLIP ViT-B/32 image tower)
  -> uint8 -> chunked RCCL all-gather; wall clock from the first H2D to the completion of the last
  gather.  value = triples/s (strong scaling: total work fixed).  --steps / --warmup count triples
  per rank are ignored: the workload is the whole set; --warmup triples are run untimed first.

Weights are the closed-form synthetic fill of oracle/synth.py (no checkpoints / network on the box);
timing is weight-independent.  Prints ONE JSON line (rank 0).  Extra objects:
  roofline      dominant MFMA kernel = the conv instantiation with the largest share of the timed
                region (labels from hf_debug_last_path): ALGORITHMIC FLOPs of its launches / their
                HIP-event durations, measured in this run on the launch stream.  peak: 157.3 TFLOP/s
                for the fp32-MFMA kernels; for the split-operand fp16 kernels (3 fp16 MFMAs per
                fp32-class product) the fp16 dense peak / 3.  `traffic` (HBM bytes per launch, PMC)
                cannot be collected inside a timing run: it is replayed from the committed rocprofv3
                --pmc pass named in `traffic_source` (null when that pass does not list the kernel).
  roofline_hbm  the same for the dominant HBM-bound (streaming) kernel: algorithmic bytes / duration
                against 8 TB/s.
  exact_f32     the generator workload with every conv on the fp32 MFMA, timed after the headline run.
  cpu_baseline  the CPU oracle (bit-identical restatement of the reference's PyTorch CPU path) timed
                on this host: batch-1 forwards with the best thread count (value) and with ONE thread
                (per_core), host core count stated (rank 0, N=1 only).
For N>1 (one rank per GPU over RCCL) the generator workload weak-scales (same per-GPU batch, uint8 result
images all-gathered asynchronously per step).  `python bench.py --gpus N` without a torchrun environment
re-executes itself as `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
--master-port <free> bench.py ...` (and refuses when fewer than N GPUs are visible); under torchrun it uses the
environment it finds and refuses a WORLD_SIZE that differs from --gpus.  The JSON line carries `ranks_observed`
(an all-reduce of ones over the process group).
--workload launch-check: the launcher / rank plumbing only (process group, barrier, max-over-ranks clock, the
  all-gather of a small uint8 tensor per step) - runs under gloo without a GPU; tests/test_parallel_gloo.py
  uses it to check that `bench.py --gpus 2` really is two ranks.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

F16_CORE_KERNELS = ("conv_mfma_h", "conv_enc_h", "gemm_h", "conv_rows_h", "stem 7x7")  # label prefixes of the fp16-MFMA kernels
PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CU x 2.4 GHz
PEAK_F16_MFMA_TFLOPS = 2516.6  # v_mfma_f32_32x32x16_f16: 32768 FLOP / 32 cycles / SIMD, 1024 SIMDs x 2.4 GHz
PEAK_HBM_GBPS = 8000.0         # MI355X_MICROARCH.md: HBM3E spec (~6.3 TB/s achievable)
# What a loop of NOTHING BUT back-to-back v_mfma_f32_32x32x16_f16 (the conv kernels' 2 x 2 tiles, three MFMAs per tile and tap,
# operands constant in registers) sustains on this chip: every SIMD issues one MFMA per 32.0 shader cycles (s_memtime) - the pipe
# is saturated - but the chip clocks at ~1.7 GHz under that load (DVFS: power), not 2.4.  tools/probes/mfma_rate.hip, run on the
# GPU box: profiles/r05_mfma_rate.txt (1788 TFLOP/s MFMA only, 1651-1661 with the kernels' 8 ds_read_b128 per 12 MFMAs);
# round 6, with the clock and the socket power sampled beside it (profiles/r06_clock_power.txt): 1799 TFLOP/s at 1.77 GHz / 1.29 kW of a
# 1.4 kW cap with non-zero operands, 2481 TFLOP/s at 2.39 GHz / 0.88 kW with zero operands - the guide's 2495 is the latter.
# `roofline.frac` stays against the nominal peak (the contract); `frac_of_sustained_mfma` is the same number against this one.
SUSTAINED_F16_MFMA_TFLOPS = 1788.0
DTYPES = {
    "f16x3": "f32 tensors + f32 accumulate; conv products as 3 fp16 MFMAs on (hi,lo)-split operands (fp32-class)",
    "f32": "f32 (fp32 MFMA)",
    "f16": "f32 tensors + f32 accumulate; conv operands rounded to fp16 (fp16 MFMA)",
}
GFLOP_PER_IMAGE = 148.52       # modulated-conv FLOPs of one 0->8 forward
GFLOP_PER_TRIPLE = 1545.0      # hot-path FLOPs of one swap (encoders + generator calls)
GFLOP_POSTPROCESS = 774.0      # PostProcessModel (594 + 2 x 90 GFLOP)
PMC_PROFILE = os.path.join("profiles", "pmc_traffic.json")
PMC_PROFILE_SWAP = os.path.join("profiles", "pmc_traffic_swap.json")  # the batched swap pass (tools/make_pmc_traffic.py --between)


def emit(out):
    """Rank 0's ONE JSON line, as the LAST line of stdout: RCCL prints a version banner through C stdio (buffered until
    exit, i.e. it would land behind the line) - flush C stdio first."""
    import ctypes

    try:
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass
    sys.stdout.flush()
    print(json.dumps(out), flush=True)


def summary_tail(out):
    """The few scalars of the line that matter, as its last object (< 600 bytes): headline, the swap half of the metric, single-swap
    latency, the fp16 / exact-fp32 runs and the roofline fractions of the three dominant kernels."""
    def g(d, *keys):
        for k in keys:
            d = d.get(k) if isinstance(d, dict) else None
        return d
    sp = out.get("swap_pipeline") or {}
    return {"generator_images_per_s": out.get("value"), "ms_per_step": out.get("ms_per_step"),
            "swap_triples_per_s": sp.get("value"), "single_swap_ms": g(sp, "single_swap", "ms_per_swap"),
            "single_swap_graph_ms": g(sp, "single_swap_graph", "ms_per_swap"),
            "f16_b16_images_per_s": g(out, "f16_mode", "value"), "exact_f32_images_per_s": g(out, "exact_f32", "value"),
            "frac_generator_dominant": g(out, "roofline", "frac"), "frac_swap_dominant": g(sp, "roofline", "frac"),
            "frac_f16_dominant": g(out, "f16_mode", "roofline", "frac"), "frac_exact_f32_dominant": g(out, "exact_f32", "roofline", "frac"),
            "mfma_busy_pmc_generator": g(out, "roofline", "mfma_busy_pmc"), "mfma_busy_pmc_swap": g(sp, "roofline", "mfma_busy_pmc"),
            "cpu_images_per_s": g(out, "cpu_baseline", "value"), "swap_verified": g(sp, "verified", "equal")}


def synth_state(prefix, shapes):
    import numpy as np

    from oracle import synth  # closed-form parameter fill only (data, not compute)

    return {k: torch.from_numpy(np.ascontiguousarray(synth.fill_value(f"{prefix}.{k}" if prefix else k, tuple(s))))
            for k, s in shapes.items()}


def build_generator(dev):
    import numpy as np

    from hairfastgan_amd.stylegan2.model import Generator
    from oracle import synth

    g = Generator(1024, 512, 8, channel_multiplier=2).eval()
    sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in
          synth.fill_state_dict({k: tuple(v.shape) for k, v in g.state_dict().items()}).items()}
    g.load_state_dict(sd)
    return g.to(dev), sd


def encoder_states():
    """Synthetic e4e / FS-encoder state dicts (shared by HairFast and the swap's CPU baseline)."""
    from oracle import ref_encoders as E

    return {"e4e": synth_state("e4e", E.e4e_param_shapes()), "fs": synth_state("fs", E.fs_param_shapes())}


def build_hairfast(sd, dev, enc=None):
    """HairFast(args) on synthetic weights; no stand-in stages (the default `Stages` raise if anything were missing)."""
    from hairfastgan_amd.hair_swap import HairFast, get_parser
    from oracle import cases as C
    from oracle import ref_postprocess as PP

    args = get_parser().parse_args([])
    args.device = dev
    pp_shapes = PP.post_process_param_shapes()
    pp_shapes.pop("latent_avg")
    enc = enc or encoder_states()
    # every network of a swap runs natively: RotateModel, ClipBlendingModel with its CLIP
    # ViT-B/32 image tower, the CtrlHair shape adaptor, SEAN
    return HairFast(args, generator_state={"g_ema": sd, "latent_avg": torch.zeros(512)},
                    e4e_state=enc["e4e"], fs_state=enc["fs"],
                    e4e_latent_avg=torch.zeros(18, 512), fs_dlatent_avg=torch.zeros(18, 512), pp_latent_avg=torch.zeros(18, 512),
                    pp_state=synth_state("pp", pp_shapes), bisenet_state=C.bisenet_params(),
                    rotate_state=synth_state("rotate", PP.rotate_param_shapes()),
                    blend_state=synth_state("clipblend", PP.clip_blending_param_shapes()), clip_state=C.clip_params(),
                    shape_state=C.shape_adaptor_params(), sean_state=C.sean_params(), sean_mean_codes=C.sean_mean_codes())


def cpu_baseline(sd, budget_s=30.0):
    """Oracle forward (generator 0->8, batch 1, explicit noise) timed on this host with 32, 64, 128 and ALL
    hardware threads (all host cores, stated; the fastest is `value`, every count tried is listed) and
    with ONE thread (`per_core`, a single forward: ~10-20 s).  Bounded: thread counts are tried in ascending
    order until the budget is spent."""
    from oracle import cases as C
    from oracle import ref_stylegan2 as O

    ncpu = os.cpu_count() or 1
    lat, nz, _ = C.generator_inputs(1024, 1, 0)
    tried = {}
    t_start = time.time()
    counts = {c for c in (32, 64, 128) if c <= ncpu} | {ncpu}
    for threads in sorted(counts):
        if tried and (time.time() - t_start) > budget_s:
            break  # bounded sample: the remaining (larger) thread counts are listed as not tried
        torch.set_num_threads(threads)
        times = []
        with torch.inference_mode():
            t0 = time.time()
            O.generator_forward(sd, lat, nz)  # warm-up
            if time.time() - t0 > 5.0:        # an oversubscribed thread count (all 256 hardware threads: 40 s per forward):
                times.append(time.time() - t0)  # its one forward is the sample
            while len(times) < 3 and (not times or (time.time() - t_start) < budget_s):
                if times and times[0] > 5.0:
                    break
                t0 = time.time()
                O.generator_forward(sd, lat, nz)
                times.append(time.time() - t0)
        times.sort()
        tried[threads] = (times[len(times) // 2], len(times))
    cores = min(tried, key=lambda k: tried[k][0])
    med, n = tried[cores]
    per_core = None
    if med * cores < 90.0:  # predicted single-thread time: keep the default run within minutes
        torch.set_num_threads(1)
        with torch.inference_mode():
            t0 = time.time()
            O.generator_forward(sd, lat, nz)
            t1 = time.time() - t0
        per_core = {"value": round(1.0 / t1, 5), "unit": "images/s", "cores": 1, "ms_per_image": round(t1 * 1e3, 1),
                    "sample": "one timed forward, torch.set_num_threads(1), no warm-up"}
    torch.set_num_threads(min(ncpu, 32))
    return {"value": round(1.0 / med, 4), "unit": "images/s", "cores": cores, "host_cores": ncpu, "kind": "port",
            "ms_per_image": round(med * 1e3, 1), "per_core": per_core,
            "sample": f"oracle (CPU restatement, bit-identical to the reference's PyTorch CPU path) generator "
                      f"0->8, batch 1, explicit noise; median of {n} timed forwards after 1 warm-up, torch "
                      f"{torch.__version__}; thread counts tried: "
                      + ", ".join(f"{k}T={v[0] * 1e3:.0f}ms" for k, v in sorted(tried.items()))
                      + "".join(f", {k}T=not tried (budget)" for k in sorted(counts) if k not in tried)}


def swap_cpu_baseline(sd, enc, threads):
    """SURVEY section 8d config 3 on the host: the hot-path call schedule of ONE swap through the CPU oracle (bit-identical
    restatement of the reference's PyTorch CPU path) - e4e B=3, FS encoder B=3, generator 3->3 B=3, 0->3 B=3, 0->8 B=1 twice
    (the reference's two Alignment.py:63 forwards), e4e B=2, generator 0->3 B=2, 4->8 B=1, 5->8 B=1: 1545 GFLOP - timed
    once at the thread count the generator baseline found fastest.  The networks between those calls (SEAN, CLIP, BiSeNet,
    shape adaptor, PostProcess) are NOT in this figure: it bounds the reference's CPU swap from below."""
    from oracle import cases as C
    from oracle import ref_encoders as E
    from oracle import ref_stylegan2 as O

    torch.set_num_threads(threads)
    gen = torch.Generator().manual_seed(3407)
    rn = lambda *s_: torch.randn(*s_, generator=gen)  # noqa: E731
    _, nz, _ = C.generator_inputs(1024, 1, 0)
    stages = {}

    def timed(name, fn):
        t0 = time.time()
        r = fn()
        stages[name] = round(stages.get(name, 0.0) + time.time() - t0, 3)
        return r

    totals = []
    with torch.inference_mode():
        for _pass in ("warm-up", "timed"):  # the first pass pays oneDNN primitive creation and the thread pool's start
            stages.clear()
            t_all = time.time()
            x256, img = rn(3, 3, 256, 256) * 0.5, rn(3, 3, 1024, 1024) * 0.5
            w = timed("e4e B=3", lambda: E.e4e_forward(enc["e4e"], x256))
            s_, fea = timed("FS encoder B=3", lambda: E.fs_encoder_test(enc["fs"], img, torch.zeros(18, 512)))
            timed("generator 3->3 B=3", lambda: O.generator_forward(sd, s_, nz, layer_in=fea, start_layer=3, end_layer=3))
            timed("generator 0->3 B=3", lambda: O.generator_forward(sd, w, nz, start_layer=0, end_layer=3))
            for _ in range(2):
                timed("generator 0->8 B=1 x2", lambda: O.generator_forward(sd, w[:1], nz))
            w2 = timed("e4e B=2", lambda: E.e4e_forward(enc["e4e"], x256[:2]))
            timed("generator 0->3 B=2", lambda: O.generator_forward(sd, w2, nz, start_layer=0, end_layer=3))
            timed("generator 4->8 B=1", lambda: O.generator_forward(sd, w[:1], nz, layer_in=rn(1, 512, 32, 32), start_layer=4, end_layer=8))
            timed("generator 5->8 B=1", lambda: O.generator_forward(sd, w[:1], nz, layer_in=rn(1, 512, 64, 64), start_layer=5, end_layer=8))
            totals.append(time.time() - t_all)
    total = totals[-1]
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    return {"value": round(1.0 / total, 4), "unit": "triples/s", "cores": threads, "host_cores": os.cpu_count(), "kind": "port",
            "s_per_triple": round(total, 2), "s_warmup_pass": round(totals[0], 2), "stage_s": stages, "gflop_per_triple": GFLOP_PER_TRIPLE,
            "sample": "ONE timed pass after one untimed warm-up pass of the hot-path call schedule of one swap (SURVEY section 8d config 3: encoders + generator "
                      f"calls, 1545 GFLOP; SEAN / CLIP / BiSeNet / shape adaptor / PostProcess excluded) through the CPU oracle, "
                      f"torch.set_num_threads({threads}) = the generator baseline's fastest count"}


def pmc_profile(kernel):
    """(bytes per launch, MFMA-busy fraction, tag) of `kernel` from the committed rocprofv3 --pmc passes
    (FETCH_SIZE / WRITE_SIZE / SQ_VALU_MFMA_BUSY_CYCLES in separate runs of this bench), or Nones."""
    try:
        with open(os.path.join(ROOT, PMC_PROFILE)) as f:
            doc = json.load(f)
        k = doc["kernels"].get(kernel)
        if k is None:
            return None, None, doc.get("tag", "r01h")
        return round(k["fetch_bytes_per_launch"] + k["write_bytes_per_launch"]), k.get("mfma_busy"), doc.get("tag", "r01h")
    except (OSError, KeyError, ValueError):
        return None, None, None


def attach_swap_pmc(rep, swap_batch, precision):
    """Fill `traffic` / `mfma_busy_pmc` of a batched swap pass's roofline from the committed rocprofv3 --pmc passes of
    `bench.py --workload swap256` (profiles/pmc_traffic_swap.json, cut to the timed region by the profile markers) - replayed,
    not measured in this run, and only when that profile was taken at the same pass size and conv precision."""
    roof = rep.get("roofline")
    if not roof:
        return
    try:
        with open(os.path.join(ROOT, PMC_PROFILE_SWAP)) as f:
            doc = json.load(f)
    except (OSError, ValueError):
        return
    if doc.get("swap_batch") != swap_batch or doc.get("conv_precision", "f16x3") != precision:
        roof["traffic_source"] = (f"{PMC_PROFILE_SWAP} holds a {doc.get('swap_batch')}-per-pass {doc.get('conv_precision', 'f16x3')} "
                                  f"profile, this run is {swap_batch}-per-pass {precision}: not attached")
        return
    k = doc.get("kernels", {}).get(roof["kernel"].replace(",split-out", ""))  # one kernel, with or without the split hand-over output
    if k is None:
        return
    traffic = round(k["fetch_bytes_per_launch"] + k["write_bytes_per_launch"])
    src = f"{PMC_PROFILE_SWAP} (tag {doc.get('tag')}): rocprofv3 --pmc passes of bench.py --workload swap256, timed region only, replayed - NOT measured in this run"
    roof.update({"traffic": traffic, "traffic_source": src, "mfma_busy_pmc": k.get("mfma_busy"),
                 "mfma_busy_pmc_source": src if k.get("mfma_busy") is not None else None,
                 "pmc_launches_averaged": k.get("launches"), "pmc_avg_launch_us_kernel_trace": k.get("avg_us")})
    if roof.get("algorithmic_bytes_per_launch"):
        roof["traffic_over_algorithmic"] = round(traffic / roof["algorithmic_bytes_per_launch"], 3)


def swap_schedule_bench(g, sd, dev, n_triples):
    """Seconds for `n_triples` replays of the per-triple hot-path schedule (after one warm-up)."""
    from hairfastgan_amd.hair_swap import HairFastHotPath, get_parser
    from oracle import ref_encoders as E

    args = get_parser().parse_args([])
    args.device = dev
    hp = HairFastHotPath(args, {"g_ema": sd, "latent_avg": torch.zeros(512)}, synth_state("e4e", E.e4e_param_shapes()),
                         synth_state("fs", E.fs_param_shapes()), torch.zeros(18, 512), torch.zeros(18, 512))
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


def make_triple_loader(n_pool=8):
    """Host-side synthetic triples: uint8 [3,1024,1024] images in pinned memory, seeded per image
    (seeds 3i, 3i+1, 3i+2 of triple i; a pool of n_pool distinct triples is cycled so that the host
    holds 75 MB instead of 2.4 GB - every triple still crosses PCIe)."""
    pool = []
    for t in range(n_pool):
        imgs = []
        for k in range(3):
            gen = torch.Generator().manual_seed(3 * t + k)
            im = torch.randint(0, 256, (3, 1024, 1024), dtype=torch.uint8, generator=gen)
            imgs.append(im.pin_memory() if torch.cuda.is_available() else im)
        pool.append(tuple(imgs))
    return lambda i: pool[i % n_pool]


def verify_gathered(hf, images, load, dev, n_total, rank, world, pass_size, groups=(0, -1)):
    """Outside the timed region: the uint8 images `parallel.swap_many` returned for this rank's first / last batched pass
    against a DIRECT `HairFast.swap_batch` call on the same triples (same seed -> same noise draws): equal bit for bit, or
    the line says by how much not.  The comparison with single `HairFast.swap` calls, stage by stage, at this pass size is
    tests/test_gpu_schedule.py::test_swap_batch_at_the_timed_pass_size_equals_single_swaps."""
    from hairfastgan_amd import parallel

    lo, hi = parallel.shard_range(n_total, rank, world)
    starts = list(range(lo, hi, pass_size))
    picked = sorted({starts[g] for g in groups if starts})
    worst, n_cmp, unequal = 0, 0, 0
    with torch.inference_mode():
        for g0 in picked:
            idx = list(range(g0, min(g0 + pass_size, hi)))
            trip = [tuple(t.to(dev) for t in load(i)) for i in idx]
            direct = hf.swap_batch(trip) if len(trip) > 1 and pass_size > 1 else [hf.swap(*tr) for tr in trip]
            for i, img in zip(idx, direct):
                ref = parallel.to_uint8_image(img * 2.0 - 1.0)
                d = int((ref.to(torch.int16) - images[i].to(ref.device).to(torch.int16)).abs().max())
                worst, n_cmp, unequal = max(worst, d), n_cmp + 1, unequal + (d != 0)
            del trip, direct
    return {"passes_checked_first_triple": picked, "images_compared": n_cmp, "images_differing": unequal,
            "max_abs_diff_uint8_levels": worst, "equal": unequal == 0,
            "what": "gathered uint8 images of the timed swap_many call vs direct HairFast.swap_batch calls on the same triples "
                    f"({pass_size} per pass), outside the timed region"}


def auto_swap_batch(n_total, world, cap=32):
    """Triples per batched pass when --swap-batch 0: the rank's whole shard, capped at the throughput plateau (32).  Measured on one
    rank's 32-triple shard with RCCL initialised (tools/probes/rank_pass_sizes.py): one pass of 32 = 443.6 ms,
    two passes of 16 = 457.3 ms - what the second pass's copy-in / the first pass's gather would hide (~5 ms) is less than what the
    smaller batch loses (13.7 ms)."""
    per_rank = -(-n_total // world)
    return max(1, min(cap, per_rank))


def kernel_report(prof, elapsed, precision, sampled=1.0, pmc=True):
    """Aggregate the HIP-event brackets of the timed region: per-kernel table + the two rooflines.
    sampled: fraction of the region's steps whose launches were bracketed (shares are scaled by it)."""
    elapsed = elapsed * sampled
    agg = {}
    for label, flops, e0, e1, nbytes in prof:
        a = agg.setdefault(label, [0.0, 0.0, 0, 0.0])
        a[0] += flops
        a[1] += e0.elapsed_time(e1) * 1e-3
        a[2] += 1
        a[3] += nbytes
    fams = {}
    for k, v in agg.items():
        row = {"avg_launch_ms": round(v[1] / v[2] * 1e3, 4), "launches": v[2], "share_of_timed_region": round(v[1] / elapsed, 4)}
        if v[0] > 0:
            row["tflops"] = round(v[0] / v[1] / 1e12, 2)
        if v[3] > 0:
            row["hbm_gbps_algorithmic"] = round(v[3] / v[1] / 1e9, 1)
        fams[k] = row
    out = {"kernels": fams}
    mfma = {k: v for k, v in agg.items() if v[0] > 0}
    if mfma:
        dom = max(mfma, key=lambda k: mfma[k][1])
        d = mfma[dom]
        ach = d[0] / d[1] / 1e12
        sustained = None
        if dom.startswith(F16_CORE_KERNELS):
            terms = 3 if precision == "f16x3" else 1
            peak = PEAK_F16_MFMA_TFLOPS / terms
            sustained = SUSTAINED_F16_MFMA_TFLOPS / terms
            peak_note = (f"fp16 dense MFMA peak {PEAK_F16_MFMA_TFLOPS} TFLOP/s / {terms} MFMA per product; "
                         f"MFMA FLOPs issued = {terms} x algorithmic = {round(ach * terms, 1)} TFLOP/s; an MFMA-only loop sustains "
                         f"{SUSTAINED_F16_MFMA_TFLOPS} TFLOP/s on this chip (pipe saturated at 32 cycles per MFMA, clock ~1.7 GHz under "
                         f"load - power-limited, 1.29 kW of a 1.4 kW cap: profiles/r05_mfma_rate.txt, r06_clock_power.txt) = {round(sustained, 1)} for this operand mode")
        else:
            peak, peak_note = PEAK_FP32_MFMA_TFLOPS, "fp32 MFMA peak"
        # the committed PMC passes were collected in the headline configuration (f16x3, batch 8, generator workload) only
        traffic, busy, tag = pmc_profile(dom) if pmc else (None, None, None)
        alg_bytes = d[3] / d[2] if d[3] > 0 else None
        out["roofline"] = {"bound": "mfma", "kernel": dom, "achieved": round(ach, 2), "peak": round(peak, 1),
                           "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                           "frac_of_sustained_mfma": None if sustained is None else round(ach / sustained, 4), "traffic": traffic,
                           "algorithmic_bytes_per_launch": None if alg_bytes is None else round(alg_bytes),
                           "traffic_over_algorithmic": None if (traffic is None or not alg_bytes) else round(traffic / alg_bytes, 3),
                           "traffic_source": (f"{PMC_PROFILE} (tag {tag}): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                                              f"bench, replayed - NOT measured in this run") if traffic is not None else None,
                           "avg_launch_ms": round(d[1] / d[2] * 1e3, 4), "launches": d[2],
                           "flops_per_launch_avg": d[0] / d[2], "peak_note": peak_note,
                           "mfma_busy_pmc": busy, "mfma_busy_pmc_source": f"{PMC_PROFILE} (tag {tag})" if busy is not None else None}
    # every MFMA kernel family of the timed region against its own roof (progress on the non-dominant ones - the fused
    # upsampling StyledConv, the 1024^2 layer, the 4^2-32^2 tower - shows here)
    fam_roof = {}
    for k, v in sorted(mfma.items(), key=lambda kv: -kv[1][1]):
        if k.startswith(F16_CORE_KERNELS):
            terms = 3 if precision == "f16x3" else 1
            pk = PEAK_F16_MFMA_TFLOPS / terms
        else:
            pk = PEAK_FP32_MFMA_TFLOPS
        a = v[0] / v[1] / 1e12
        fam_roof[k] = {"achieved": round(a, 2), "peak": round(pk, 1), "frac": round(a / pk, 4),
                       "frac_of_sustained_mfma": round(a / (pk * SUSTAINED_F16_MFMA_TFLOPS / PEAK_F16_MFMA_TFLOPS), 4) if k.startswith(F16_CORE_KERNELS) else None,
                       "avg_launch_ms": round(v[1] / v[2] * 1e3, 4),
                       "launches": v[2], "share_of_timed_region": round(v[1] / elapsed, 4)}
    if fam_roof:
        out["roofline_families"] = {"bound": "mfma", "unit": "TFLOP/s", "kernels": fam_roof}
    hbm = {k: v for k, v in agg.items() if v[3] > 0 and v[0] == 0}  # streaming kernels (no MFMA work)
    if hbm:
        dom = max(hbm, key=lambda k: hbm[k][1])
        d = hbm[dom]
        ach = d[3] / d[1] / 1e9
        traffic, _, tag = pmc_profile(dom) if pmc else (None, None, None)
        out["roofline_hbm"] = {"bound": "hbm", "kernel": dom, "achieved": round(ach, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                               "frac": round(ach / PEAK_HBM_GBPS, 4), "traffic": traffic,
                               "traffic_source": f"{PMC_PROFILE} (tag {tag}), replayed" if traffic is not None else None,
                               "avg_launch_ms": round(d[1] / d[2] * 1e3, 4), "launches": d[2],
                               "bytes_per_launch_avg": d[3] / d[2]}
    return out


def balance_report(st, cpu_slice):
    """Per-rank imbalance and the exposed tail of the all-gather of one parallel.swap_many call (rank 0's view)."""
    if not st:
        return None
    return {"per_rank_compute_s": [round(v, 4) for v in st["per_rank_compute_s"]], "imbalance_max_over_min": round(st["imbalance"], 4),
            "gather_tail_s_rank0": round(st["gather_tail_s"], 4),
            "cpu_threads_rank0": None if cpu_slice is None else f"{cpu_slice[0]}-{cpu_slice[-1]} ({len(cpu_slice)} pinned)"}


def self_launch(args, argv):
    """`python bench.py --gpus N` (N > 1) outside a torchrun environment: become the launcher of N ranks."""
    import socket
    import subprocess

    if args.workload != "launch-check":
        n_vis = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if n_vis < args.gpus:
            sys.exit(f"bench.py: --gpus {args.gpus} but only {n_vis} GPU(s) visible - refusing to report fewer ranks than asked for")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
    sys.exit(subprocess.call(cmd, env=env))


def launch_check(args):
    """The N-rank plumbing of this file without the GPU workload (gloo on CPU, RCCL when GPUs are there)."""
    import torch.distributed as dist

    from hairfastgan_amd import parallel

    rank, world, local = parallel.init_from_env()
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    cuda = torch.cuda.is_available()
    dev = torch.device("cuda", local) if cuda else torch.device("cpu")
    ones = torch.ones(1, device=dev)
    if world > 1:
        dist.all_reduce(ones)
        dist.barrier()
    t0 = time.perf_counter()
    got = None
    for k in range(args.steps):
        u8 = torch.full((2, 3, 8, 8), (rank * 16 + k) % 256, dtype=torch.uint8, device=dev)
        got = parallel.all_gather_images(u8)
    if world > 1:
        dist.barrier()
    sec = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(sec, op=dist.ReduceOp.MAX)
    ok = got.shape[0] == 2 * world and all(int(got[2 * r, 0, 0, 0]) == (r * 16 + args.steps - 1) % 256 for r in range(world))
    # the rest of what an N-rank bench run does around its GPU work, at THIS rank count: the per-rank core slice
    # (len(cpus) // local world) and a ragged block partition through parallel.swap_many (3 * world + 5 triples: shards of
    # different lengths, padded gather rounds, results in triple order on every rank) - with a stand-in for HairFast.swap
    cpu_slice = parallel.pin_rank_to_cores(local, int(os.environ.get("LOCAL_WORLD_SIZE", world)))
    n_tr = 3 * world + 5
    load = lambda i: tuple(torch.full((3, 4, 4), (i * 3 + j) % 251, dtype=torch.uint8) for j in range(3))  # noqa: E731
    fake_swap = lambda a, b, c: (a.float() + b.float() + c.float()) / 765.0  # noqa: E731
    st = {}
    imgs, n_local = parallel.swap_many(fake_swap, n_tr, load, device=dev if cuda else None, chunk=2, batch=2,
                                       swap_batch_fn=lambda trs: [fake_swap(*t) for t in trs], stats=st)
    want = torch.stack([parallel.to_uint8_image(fake_swap(*load(i)) * 2.0 - 1.0) for i in range(n_tr)])
    ragged_ok = tuple(imgs.shape) == (n_tr, 3, 4, 4) and bool(torch.equal(imgs.cpu(), want))
    slices = [None] * world
    if world > 1:
        dist.all_gather_object(slices, (n_local, None if cpu_slice is None else (cpu_slice[0], cpu_slice[-1], len(cpu_slice))))
    else:
        slices = [(n_local, None)]
    shards = [s[0] for s in slices]
    cores = [s[1] for s in slices]
    disjoint = all(c is None for c in cores) or all(cores[i][1] < cores[i + 1][0] for i in range(world - 1) if cores[i] and cores[i + 1])
    ok = ok and ragged_ok and sum(shards) == n_tr and disjoint
    if rank == 0:
        print(json.dumps({"metric": "launch_check", "value": int(ones.item()), "unit": "ranks", "n_gpus": world,
                          "ranks_observed": int(ones.item()), "steps": args.steps, "warmup": 0,
                          "ms_per_step": round(float(sec.item()) / max(args.steps, 1) * 1e3, 4), "gather_ok": bool(ok),
                          "ragged_swap_many_ok": bool(ragged_ok), "triples": n_tr, "shards": shards,
                          "core_slices_first_last_count": cores, "core_slices_disjoint": bool(disjoint),
                          "backend": dist.get_backend() if world > 1 else None, "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
                          "config": {"workload": "launcher / process-group check (no GPU work)"}}), flush=True)
    if world > 1:
        dist.destroy_process_group()
    if not ok:
        sys.exit(4)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=8, help="images per GPU per step (generator workload)")
    ap.add_argument("--workload", choices=("generator", "swap256", "launch-check"), default="generator")
    ap.add_argument("--triples", type=int, default=256, help="swap256: triples of the whole job")
    ap.add_argument("--swap-batch", type=int, default=0,
                    help="swap256: triples per batched pass over the hot path (HairFast.swap_batch); 1 = one HairFast.swap per triple; "
                         "0 (default) = auto: min(32, the rank's shard) - one pass of 32 beats two of 16 even with nothing to overlap "
                         "its copy-in and gather with (DESIGN.md section 7). "
                         "Measured on resident inputs (tools/probes/swap_batch_sizes.py, round 4): 1: 29 triples/s, 4: 54, 8: 64, "
                         "16: 70.6 (22.6 GiB), 32: 72.9 (32.1 GiB), 48: 72.4, 64: 72.8 (50.9 GiB) - a plateau from 32, which is "
                         "also one pass per rank for 256 triples on 8 GPUs")
    ap.add_argument("--precision", choices=("f16x3", "f32", "f16"), default=None,
                    help="matrix-core mode of the 3x3 convs (default: HAIRFAST_CONV_PRECISION or f16x3)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-exact-f32", action="store_true", help="skip the fp32-MFMA comparison run (profiling)")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--no-verify", action="store_true", help="swap workloads: skip the comparison of gathered images with direct swap_batch calls")
    ap.add_argument("--event-every", type=int, default=10,
                    help="generator workload: bracket the launches of every N-th timed step (from step N // 2) with HIP events (1 = every step)")
    ap.add_argument("--priming", type=int, default=6,
                    help="generator workload: untimed forwards BEFORE the --warmup steps (plans, allocator and clocks settle over the first ~4 "
                         "forwards of a process: tools/probes/step_ramp.py); reported as priming_steps")
    ap.add_argument("--pipeline-triples", type=int, default=256,
                    help="generator workload: triples of the WHOLE job for the secondary swap_pipeline object (strong scaling over --gpus)")
    ap.add_argument("--swap-triples", type=int, default=4,
                    help="generator workload: triples per GPU for the secondary hair-swap measurements (0 = skip)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        self_launch(args, sys.argv[1:])
    if args.workload == "launch-check":
        return launch_check(args)

    from hairfastgan_amd import _marshal, _runtime, parallel

    swap_batch_auto = args.swap_batch <= 0
    if args.precision:
        _runtime.set_conv_precision(args.precision)
    precision = _runtime.conv_precision()

    rank, world, local = parallel.init_from_env()
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world} - refusing to print a line for a different rank count")
    # one node: every rank gets its own slice of the host's hardware threads (launch-bound ranks otherwise share cores)
    cpu_slice = parallel.pin_rank_to_cores(local, int(os.environ.get("LOCAL_WORLD_SIZE", world)))
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    if torch.cuda.device_count() <= local:
        sys.exit(f"bench.py: rank {rank} (local {local}) has no GPU: {torch.cuda.device_count()} visible")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    import torch.distributed as dist

    use_dist = world > 1 or os.environ.get("HF_FORCE_DIST", "0") == "1"

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(sec):
        if not use_dist:
            return sec
        t = torch.tensor([sec], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    ranks_observed = 1
    if use_dist:
        t_ranks = torch.ones(1, device=dev)
        dist.all_reduce(t_ranks)
        ranks_observed = int(t_ranks.item())
        assert ranks_observed == world, f"process group reports {ranks_observed} ranks, WORLD_SIZE {world}"

    g, sd = build_generator(dev)

    # =========================== workload swap256 (BASELINE.json configs[3]) ===========================
    if args.workload == "swap256":
        if swap_batch_auto:
            args.swap_batch = auto_swap_batch(args.triples, world)
        hf = build_hairfast(sd, dev)
        load = make_triple_loader()
        with torch.inference_mode():
            for i in range(max(1, args.warmup)):
                hf.swap(*[t.to(dev) for t in load(i)])
            if args.swap_batch > 1:  # warm the batched shapes too (weight splits, plans)
                hf.swap_batch([tuple(t.to(dev) for t in load(i)) for i in range(args.swap_batch)])
        barrier()
        prof = None if args.no_kernel_events else []
        _marshal.PROFILE = prof
        # hf_profile_marker_kernel dispatches bracket the timed region: tools/summarize_prof.py / make_pmc_traffic.py cut
        # rocprofv3's per-dispatch tables to it (warm-up and plan-time kernels - weight splits, conv_prepare - excluded)
        _runtime.lib().hf_profile_marker(1, _runtime.stream())
        t0 = time.perf_counter()
        st256 = {}
        images, n_local = parallel.swap_many(lambda a, b, c: hf.swap(a, b, c), args.triples, load, device=dev,
                                             batch=args.swap_batch, swap_batch_fn=hf.swap_batch if args.swap_batch > 1 else None,
                                             stats=st256)
        barrier()
        elapsed = max_over_ranks(time.perf_counter() - t0)
        _runtime.lib().hf_profile_marker(2, _runtime.stream())
        _marshal.PROFILE = None
        assert images.shape == (args.triples, 3, 1024, 1024) and images.dtype == torch.uint8
        verified = None if args.no_verify else verify_gathered(hf, images, load, dev, args.triples, rank, world, args.swap_batch)
        if rank == 0:
            out = {"metric": "hair_swap_triples_per_sec", "value": round(args.triples / elapsed, 3), "unit": "triples/s",
                   "n_gpus": world, "ranks_observed": ranks_observed, "steps": args.triples, "warmup": max(1, args.warmup),
                   "ms_per_step": round(elapsed / args.triples * 1e3 * world, 4), "ms_per_triple_per_gpu": round(elapsed / max(n_local, 1) * 1e3, 3),
                   "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": DTYPES[precision],
                   "data": "synthetic",
                   "config": {"swap_batch": args.swap_batch,
                              "batching": (f"HairFast.swap_batch: {args.swap_batch} triples per pass - every hot-path call below runs once with "
                                           f"{args.swap_batch}x the listed batch" if args.swap_batch > 1 else "one HairFast.swap per triple"),
                              "workload": f"{args.triples} synthetic 1024^2 triples sharded over {world} GPU(s): host uint8 -> H2D -> "
                                          "HairFast.swap (e4e B=3, FS-encoder B=3, gen 3->3 B=3, gen 0->3 B=3, gen 0->8 B=2 [both "
                                          "Alignment rotations batched], e4e B=2, gen 0->3 B=2, gen 4->8 B=1, PostProcess encoder "
                                          "[774 GFLOP], gen 5->8 B=1, BiSeNet parsing x5, RotateModel, ClipBlendingModel incl. its CLIP ViT-B/32 image tower, CtrlHair shape adaptor, SEAN encode + 2 decodes - all native) -> uint8 -> "
                                          "chunked RCCL all-gather; wall from first H2D to last gather (BASELINE.json configs[3])",
                              "triples": args.triples, "triples_per_gpu": n_local, "parallelism": f"replica x{world}, block-partitioned triples",
                              "weights": "synthetic closed-form (oracle/synth.py)", "conv_precision": precision,
                              "gather": "RCCL all_gather_into_tensor of uint8 images per 8 local triples (async)" if use_dist else "single process: no collective"},
                   "algorithmic_tflops_hot_path": round(args.triples * (GFLOP_PER_TRIPLE + GFLOP_POSTPROCESS) / elapsed / 1e3, 2),
                   "gflop_per_triple": GFLOP_PER_TRIPLE + GFLOP_POSTPROCESS,
                   "balance": balance_report(st256, cpu_slice), "verified": verified}
            if prof:
                rep = kernel_report(prof, elapsed, precision, pmc=False)
                attach_swap_pmc(rep, args.swap_batch, precision)
                out.update(rep)
            emit(out)
        if use_dist:
            dist.barrier()
            dist.destroy_process_group()
        return

    # =========================== workload generator (BASELINE.json configs[1]) ===========================
    B = args.batch
    torch.manual_seed(3407 + rank)  # the reference's default seed (utils/seed.py:19)
    latent = torch.randn(B, 18, 512, device=dev)
    gather_note = None
    pending = None

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

    for _ in range(max(0, args.priming) + args.warmup):
        step()
    if pending is not None:
        pending[1].wait()
    barrier()
    prof = None if args.no_kernel_events else []
    every = max(1, args.event_every)
    first_ev = min(every // 2, max(0, args.steps - 1))  # (not step 0: the host is still filling the stream there)
    n_bracketed = len(range(first_ev, args.steps, every))
    # (hf_profile_marker_kernel dispatches bracket the timed region: tools/summarize_prof.py --between / make_pmc_traffic.py --between
    # cut rocprofv3's per-dispatch tables to it - the synthetic fill's copies and the one-off weight preparation stay outside)
    _runtime.lib().hf_profile_marker(1, _runtime.stream())
    t0 = time.perf_counter()
    for i in range(args.steps):
        # per-kernel HIP events (the `roofline` durations) on every `every`-th step of the timed region: an event pair costs
        # ~8 us of serialised stream time, 41 pairs per forward = 0.3 ms - bracketing every step cost 6 % of the headline
        _marshal.PROFILE = prof if (prof is not None and i >= first_ev and (i - first_ev) % every == 0) else None
        step()
    _marshal.PROFILE = None
    if pending is not None:
        pending[1].wait()
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    _runtime.lib().hf_profile_marker(2, _runtime.stream())
    overflow = _marshal.f16_overflow_count(_runtime.lib())

    # comparison runs (outside the timed region): the same forward with every conv on the exact-fp32
    # MFMA, and BASELINE.json configs[4] (fp16 operands, batch 16)
    def alt_run(mode, batch, events=None):
        prev = _runtime.set_conv_precision(mode)
        lat = torch.randn(batch, 18, 512, device=dev)
        try:
            with torch.inference_mode():
                for _ in range(2):
                    g([lat], input_is_latent=True)
                torch.cuda.synchronize()
                _marshal.PROFILE = events
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    g([lat], input_is_latent=True)
                torch.cuda.synchronize()
                return time.perf_counter() - t1
        finally:
            _marshal.PROFILE = None
            _runtime.set_conv_precision(prev)

    exact_f32 = f16_mode = None
    if world == 1 and not args.no_exact_f32:
        if precision != "f32":
            ev32 = None if args.no_kernel_events else []
            e32 = alt_run("f32", B, ev32)
            exact_f32 = {"value": round(B * args.steps / e32, 3), "unit": "images/s", "ms_per_step": round(e32 / args.steps * 1e3, 4),
                         "note": "same forward, HAIRFAST_CONV_PRECISION=f32 (v_mfma_f32_32x32x2_f32 only): the fallback headline "
                                 "should the fp16 split ever clamp on a trained checkpoint"}
            if ev32:
                exact_f32["roofline"] = kernel_report(ev32, e32, "f32", pmc=False).get("roofline")
        e16 = alt_run("f16", 16)
        ev16 = None if args.no_kernel_events else []
        e16_ev = alt_run("f16", 16, ev16) if ev16 is not None else None  # a second, event-bracketed run for the rooflines only
        f16_mode = {"value": round(16 * args.steps / e16, 3), "unit": "images/s", "ms_per_step": round(e16 / args.steps * 1e3, 4),
                    "batch": 16, "note": "BASELINE.json configs[4]: fp16 conv operands (HAIRFAST_CONV_PRECISION=f16; the hand-over "
                                         "activations between convs are fp16, everything else fp32), fp32 accumulation and "
                                         "demodulation; batch-16 rows checked against the reference goldens in tests/test_gpu_parity.py"}
        if ev16:  # configs[4] under its own roof: one fp16 MFMA per product -> the full 2516.6 TFLOP/s dense peak
            rep16 = kernel_report(ev16, e16_ev, "f16", pmc=False)
            f16_mode["roofline"], f16_mode["roofline_families"] = rep16.get("roofline"), rep16.get("roofline_families")
            f16_mode["algorithmic_tflops_whole_forward"] = round(16 * args.steps / e16 * GFLOP_PER_IMAGE / 1e3, 2)
            f16_mode["note"] += "; rooflines from a second run of the same steps with every launch bracketed by HIP events"

    # Secondary measurements (outside the timed region above) of BASELINE.json configs[2]/[3] on a bounded
    # sample: (a) the hot-path kernels of one swap replayed on resident tensors (eager / hipGraph),
    # (b) the complete HairFast.swap pipeline incl. H2D, the stages in between and the uint8 gather path.
    swap_info = pipeline_info = None
    if args.swap_triples > 0:
        swap_res = swap_schedule_bench(g, sd, dev, args.swap_triples)
        times = {m: (None if swap_res.get(m) is None else max_over_ranks(swap_res[m])) for m in ("eager", "hipgraph")}
        best = min((v for v in times.values() if v is not None), default=None)
        swap_info = {"metric": "hair_swap_hot_path_triples_per_sec",
                     "value": None if best is None else round(args.swap_triples * world / best, 3),
                     "unit": "triples/s",
                     "ms_per_triple": {m: (None if v is None else round(v / args.swap_triples * 1e3, 2)) for m, v in times.items()},
                     "errors": {k: v for k, v in swap_res.items() if k.endswith("_error")},
                     "triples_per_gpu": args.swap_triples,
                     "workload": "per triple: e4e B=3, FS-encoder B=3, gen 3->3 B=3, gen 0->3 B=3, gen 0->8 B=2 (the two Alignment.py:63 "
                                 "forwards batched), e4e B=2, gen 0->3 B=2, gen 4->8 B=1, gen 5->8 B=1 (1545 GFLOP; the reference's "
                                 "discarded FS-encoder generator forward is not run); resident synthetic tensors between the calls",
                     "gflop_per_triple": GFLOP_PER_TRIPLE}
        try:
            enc_states = encoder_states()
            hf = build_hairfast(sd, dev, enc_states)
            load = make_triple_loader(2)
            n_pipe = max(args.pipeline_triples, world)
            if swap_batch_auto:
                args.swap_batch = auto_swap_batch(n_pipe, world)
            with torch.inference_mode():
                hf.swap(*[t.to(dev) for t in load(0)])
                if args.swap_batch > 1:
                    hf.swap_batch([tuple(t.to(dev) for t in load(i)) for i in range(args.swap_batch)])
            barrier()
            # BASELINE.json configs[3] at its own size, strong-scaled: --pipeline-triples (default 256) triples of the WHOLE job,
            # block-partitioned over the ranks - the driver's plain `bench.py --gpus N` yields the triples/s curve
            st_pipe = {}
            t0 = time.perf_counter()
            pipe_images, _ = parallel.swap_many(lambda a, b, c: hf.swap(a, b, c), n_pipe, load, device=dev, batch=args.swap_batch,
                                                swap_batch_fn=hf.swap_batch if args.swap_batch > 1 else None, stats=st_pipe)
            barrier()
            tp = max_over_ranks(time.perf_counter() - t0)
            pipe_verified = None if args.no_verify else verify_gathered(hf, pipe_images, load, dev, n_pipe, rank, world, args.swap_batch)
            del pipe_images
            # the kernels of one batched pass under their roofs: every conv / GEMM launch bracketed by HIP events
            pipe_roof = None
            if not args.no_kernel_events and args.swap_batch > 1:
                trip = [tuple(t.to(dev) for t in load(i)) for i in range(args.swap_batch)]
                torch.cuda.synchronize()
                _marshal.PROFILE = evp = []
                t1 = time.perf_counter()
                with torch.inference_mode():
                    hf.swap_batch(trip)
                torch.cuda.synchronize()
                t_pass = time.perf_counter() - t1
                _marshal.PROFILE = None
                pipe_roof = kernel_report(evp, t_pass, precision, pmc=False)
                attach_swap_pmc(pipe_roof, args.swap_batch, precision)
                pipe_roof.pop("kernels", None)
                pipe_roof["pass_ms_with_events"] = round(t_pass * 1e3, 2)
                del trip
            # the same pipeline one swap at a time (the reference's own protocol, utils/time.py): latency of a single swap
            n_single = args.swap_triples * world
            t0 = time.perf_counter()
            parallel.swap_many(lambda a, b, c: hf.swap(a, b, c), n_single, load, device=dev)
            barrier()
            ts = max_over_ranks(time.perf_counter() - t0)
            # where a batched pass spends its time: event marks at the stage boundaries of one swap_batch call
            stage_ms = None
            try:
                from hairfastgan_amd import hair_swap as HS

                trip = [tuple(t.to(dev) for t in load(i)) for i in range(args.swap_batch)]
                with torch.inference_mode():
                    HS.STAGE_MARKS = marks = []
                    hf.swap_batch(trip)
                    HS.STAGE_MARKS = None
                torch.cuda.synchronize()
                stage_ms = {}
                for (_, e0), (lab, e1) in zip(marks[:-1], marks[1:]):
                    stage_ms[lab] = round(stage_ms.get(lab, 0.0) + e0.elapsed_time(e1) / max(1, args.swap_batch), 3)
            except Exception as e:
                stage_ms = {"error": f"{type(e).__name__}: {e}"[:200]}
            finally:
                try:
                    HS.STAGE_MARKS = None
                except Exception:
                    pass
            # ... and the whole swap as one hipGraph replay (HairFast.swap_graphed): the same kernels without the host's launch gaps
            graph_info = None
            try:
                dev_triples = [tuple(t.to(dev) for t in load(i)) for i in range(2)]
                with torch.inference_mode():
                    hf.swap_graphed(*dev_triples[0])
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for i in range(n_single):
                        hf.swap_graphed(*dev_triples[i % 2])
                    torch.cuda.synchronize()
                    tg = time.perf_counter() - t0
                graph_info = {"ms_per_swap": round(tg / n_single * 1e3, 2), "triples": n_single,
                              "note": "HairFast.swap_graphed: one hipGraph replay per swap, images resident on the GPU"}
            except Exception as e:
                graph_info = {"error": f"{type(e).__name__}: {e}"[:300]}
            pipeline_info = {"metric": "hair_swap_triples_per_sec", "value": round(n_pipe / tp, 3), "unit": "triples/s",
                             "ms_per_triple_per_gpu": round(tp / (n_pipe / world) * 1e3, 2), "triples": n_pipe,
                             "scaling": "strong", "swap_batch": args.swap_batch, "balance": balance_report(st_pipe, cpu_slice),
                             "verified": pipe_verified,
                             "algorithmic_tflops_hot_path": round(n_pipe * (GFLOP_PER_TRIPLE + GFLOP_POSTPROCESS) / tp / 1e3 / world, 2),
                             "single_swap": {"ms_per_swap": round(ts / (n_single / world) * 1e3, 2), "triples": n_single,
                                             "note": "one HairFast.swap per triple (no batching across triples)"},
                             "single_swap_graph": graph_info,
                             "stage_ms_per_triple": stage_ms,
                             "workload": f"python bench.py --workload swap256 --triples {n_pipe}: host uint8 -> H2D -> HairFast.swap_batch "
                                         "(every network native, no stand-ins: SEAN, CLIP ViT-B/32 tower, shape adaptor, RotateModel, "
                                         "PostProcess, BiSeNet) -> uint8 -> gather; BASELINE.json configs[3] (synthetic weights), total work fixed"}
            if pipe_roof:
                pipeline_info.update({k: pipe_roof[k] for k in ("roofline", "roofline_families", "pass_ms_with_events") if k in pipe_roof})
            pipeline_info["_enc_states"] = enc_states
        except Exception as e:
            pipeline_info = {"error": f"{type(e).__name__}: {e}"[:300]}

    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        value = B * args.steps * world / elapsed
        out = {
            "metric": "stylegan2_generator_fwd_1024_images_per_sec", "value": round(value, 3), "unit": "images/s",
            "n_gpus": world, "ranks_observed": ranks_observed, "steps": args.steps, "warmup": args.warmup, "priming_steps": max(0, args.priming), "ms_per_step": round(ms_step, 4),
            "ms_per_image": round(ms_step / B, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": DTYPES[precision], "data": "synthetic",
            "config": {"workload": "StyleGAN2 1024^2 generator forward (range 0->8), random W+, fresh noise per layer, "
                                   "HIP modulated-conv + upfirdn2d kernels (BASELINE.json configs[1])",
                       "batch_per_gpu": B, "global_batch": B * world, "parallelism": f"replica x{world}",
                       "weights": "synthetic closed-form (oracle/synth.py)", "conv_precision": precision},
            "algorithmic_tflops_whole_forward": round(value * GFLOP_PER_IMAGE / 1e3 / world, 2),
            "f16_split_clamped_elements": overflow,
            "timing_note": (f"the conv / streaming launches of every {every}th step of the timed region ({n_bracketed} of {args.steps} steps) "
                            "are bracketed by pairs of HIP events (the per-kernel durations of `roofline`); a bracketed step costs about "
                            "0.3 ms more, which the headline includes (--no-kernel-events measures without any)") if prof else "no per-kernel events in the timed region",
            "priming_note": (f"{max(0, args.priming)} untimed forwards ran before the {args.warmup} warm-up steps: the first ~4 forwards of a process are "
                             "slower (plans, allocator growth, clocks: tools/probes/step_ramp.py, 97 / 5.1 / 4.8 / 4.6 / 4.5 ms); the timed region is "
                             "still exactly --steps forwards between two synchronisations"),
        }
        if gather_note:
            out["config"]["gather"] = gather_note
        elif use_dist:
            out["config"]["gather"] = "async RCCL all_gather_into_tensor of uint8 images, one per step"
        if prof:
            out.update(kernel_report(prof, elapsed, precision, n_bracketed / args.steps))
        enc_states = pipeline_info.pop("_enc_states", None) if pipeline_info else None
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sd)
            if enc_states is not None and "error" not in pipeline_info:  # the swap metric's own CPU figure
                try:
                    pipeline_info["cpu_baseline"] = swap_cpu_baseline(sd, enc_states, out["cpu_baseline"]["cores"])
                except Exception as e:
                    pipeline_info["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"[:200]}
        if exact_f32 is not None:
            out["exact_f32"] = exact_f32
        if f16_mode is not None:
            out["f16_mode"] = f16_mode
        if swap_info is not None:
            out["swap_schedule"] = swap_info
        if pipeline_info is not None:
            out["swap_pipeline"] = pipeline_info
        out["summary"] = summary_tail(out)  # LAST key: a reader that only keeps the end of the line still sees the scalars that matter
        emit(out)

    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
