"""Device / stream plumbing shared by the public ops."""
import os
import threading

import torch

from . import _lib


def lib():
    global _lib_flag
    L = _lib.load()
    if _lib_flag != _batch_invariant:  # HAIRFAST_DETERMINISTIC / set_batch_invariant -> the library's process-wide flag
        L.hf_set_batch_invariant(1 if _batch_invariant else 0)
        _lib_flag = _batch_invariant
    return L


def require_gpu(*tensors):
    """Entry check of every public op: GPU tensors only, and no autograd - the kernels are forward
    only (the reference's ops are differentiable, op/fused_act.py:19-70, op/upfirdn2d.py:19-142;
    HairFast runs them frozen under inference_mode, models/Net.py:44-46), so a caller that expects
    gradients is told instead of silently receiving detached outputs."""
    grad = torch.is_grad_enabled()
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError(
                "hairfastgan_amd ops run on the MI355X only (got a CPU tensor); there is no CPU fallback - "
                "the CPU restatement lives in oracle/ and is test infrastructure")
        if grad and t.requires_grad and not isinstance(t, torch.nn.Parameter):  # module parameters are frozen by contract
            raise RuntimeError(
                "hairfastgan_amd is inference only: an input requires grad while autograd is enabled; "
                "wrap the call in torch.inference_mode() / torch.no_grad() or detach the input")


# In-kernel split-K reduction (csrc/conv_common.h splitk_arrive_last): the kernels count the arrivals of a tile's K-split
# blocks in a caller-owned zero-initialised buffer that every launch leaves zero again.  One buffer per (device, stream) -
# launches of one stream are ordered, launches of different streams (the Embedding stage's side streams) may overlap -
# registered with the library whenever the current stream changes.
# OFF by default (HAIRFAST_SPLITK_INKERNEL=1 / set_splitk_inkernel turn it on): measured on the MI355X it LOSES - a single
# swap 34.4 -> 57.6 ms (328 splitk_reduce launches saved, but the split-K kernels themselves 3-5x slower): the release /
# acquire fences that make one block's slab visible to a block on another XCD are L2 write-backs / invalidations
# (buffer_wbl2 / buffer_inv sc1), paid by every block of every split-K launch, and the last block's z-ordered re-read of
# all slabs is serial.  Results are bit-identical to the two-launch form (tests/test_gpu_encoders.py).
SPLITK_COUNTER_INTS = 1 << 16
_splitk_inkernel = os.environ.get("HAIRFAST_SPLITK_INKERNEL", "0") not in ("", "0")
_counter_bufs = {}
# which (device, stream) buffer the CALLING THREAD last registered: the library keeps the pointer in thread_local storage
# (hf_set_splitk_counters), so the Python-side memo is per thread as well - a second thread launching on the same stream
# registers for itself instead of silently taking the two-launch form
_counter_tls = threading.local()
_counter_epoch = [0]  # bumped by set_splitk_inkernel: every thread re-registers (or, switched off, never registers again)


def set_splitk_inkernel(on):
    """Process-wide switch of the in-kernel split-K reduction (returns the previous setting); off = every split-K launch is
    followed by the splitk_reduce kernel.  Results are bit-identical either way."""
    global _splitk_inkernel
    prev, _splitk_inkernel = _splitk_inkernel, bool(on)
    _counter_epoch[0] += 1
    _counter_tls.key = None
    if not on:  # clears the calling thread's pointer; other threads' pointers are dropped the next time they ask for a stream
        lib().hf_set_splitk_counters(None, 0)
    return prev


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_GET_DEVICE = getattr(torch._C, "_cuda_getDevice", None)
if _GET_DEVICE is None:
    _RAW_STREAM = None


def stream():
    """Raw hipStream_t of torch's current stream (kernels are enqueued asynchronously on it,
    like the reference's at::cuda::getCurrentCUDAStream(), upfirdn2d_kernel.cu:213-215)."""
    # (the raw handle straight from torch's C layer: torch.cuda.current_stream() builds a Stream object through three
    # Python layers, ~4 us - and a single swap asks 900 times while its ~2100 launches are HOST-bound: 31.2 ms of enqueueing
    # for 32.3 ms of wall, tools/probes/swap_host_profile.py)
    if _RAW_STREAM is not None:
        dev_index = _GET_DEVICE()
        raw = _RAW_STREAM(dev_index)
    else:
        s_ = torch.cuda.current_stream()
        dev_index, raw = s_.device_index, s_.cuda_stream
    if _splitk_inkernel:
        key = (dev_index, raw, _counter_epoch[0])
        if key != getattr(_counter_tls, "key", None):
            buf = _counter_bufs.get(key[:2])
            if buf is None:
                buf = _counter_bufs[key[:2]] = torch.zeros(SPLITK_COUNTER_INTS, dtype=torch.int32, device=torch.device("cuda", dev_index))
            lib().hf_set_splitk_counters(buf.data_ptr(), SPLITK_COUNTER_INTS)
            _counter_tls.key = key
    elif getattr(_counter_tls, "key", None) is not None:  # switched off by another thread: drop this thread's pointer too
        lib().hf_set_splitk_counters(None, 0)
        _counter_tls.key = None
    return raw


# ---------------------------------------------------------------------------------------
# Matrix-core mode of the 3x3 (modulated) convolutions.  Tensors are fp32 in HBM and the
# accumulation is fp32 in every mode; the mode picks the MFMA the products run on:
#   "f16x3" (default)  split-operand fp16 MFMA (csrc/convh.hip, nterms 3): each fp32 operand
#                      is carried as an fp16 (hi, lo) pair, a*b = hi*hi + hi*lo + lo*hi;
#                      products carry 22 operand bits - the result differs from the fp32 MFMA
#                      one by ~1e-6 relative, the same class as an fp32 reassociation - at
#                      ~2.7x the fp32 MFMA throughput.
#   "f32"              v_mfma_f32_32x32x2_f32 only (csrc/modconv.hip): exact fp32 products.
#   "f16"              plain fp16 operands (nterms 1): BASELINE.json configs[4]'s fp16 mode,
#                      ~5e-4 relative error per product.
# Layers whose shape the fp16 kernels do not take run the fp32 kernels in every mode.
#   "auto"             f16x3 with a safety net: the forward (Generator.forward, HairFast.swap / swap_batch) is followed by a
#                      read of the library's clamp counter (hf_f16_overflow_count: elements the fp16 split had to
#                      saturate, |s*x| > 131008 or NaN) and RE-RUN on the exact fp32 kernels when it is not zero.  Two
#                      device synchronisations per guarded call; not usable inside a hipGraph capture (there it runs
#                      unguarded f16x3).  For the first runs on a new checkpoint (tools/check_checkpoint.py).
CONV_PRECISIONS = ("f16x3", "f32", "f16", "auto")
_conv_precision = os.environ.get("HAIRFAST_CONV_PRECISION", "f16x3")
if _conv_precision not in CONV_PRECISIONS:
    raise ValueError(f"HAIRFAST_CONV_PRECISION must be one of {CONV_PRECISIONS}, got {_conv_precision!r}")
_auto_depth = 0        # > 0 inside a guarded call (nested guards do nothing)
auto_reruns = 0        # how many guarded calls were repeated in f32 (diagnostics, tests)


def conv_precision():
    """The mode the kernels run in now: "auto" is f16x3 (or f32 during a re-run)."""
    return "f16x3" if _conv_precision == "auto" else _conv_precision


def configured_conv_precision():
    return _conv_precision


def run_guarded(fn):
    """Mode "auto": fn() on the f16x3 kernels; if the fp16 split clamped anything meanwhile, fn() again on the exact
    fp32 kernels (fn must be repeatable: same inputs, explicit or re-seeded noise).  Any other mode: fn()."""
    global _conv_precision, _auto_depth, auto_reruns
    if _conv_precision != "auto" or _auto_depth > 0 or (torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()):
        return fn()
    L = lib()
    _auto_depth += 1
    try:
        L.hf_f16_overflow_count(1)          # synchronises; clears the counter
        out = fn()
        if int(L.hf_f16_overflow_count(0)) > 0:
            _conv_precision = "f32"
            try:
                out = fn()
                auto_reruns += 1
            finally:
                _conv_precision = "auto"
        return out
    finally:
        _auto_depth -= 1


def set_conv_precision(mode):
    """Process-wide; returns the previous mode.  Modules cache nothing mode-specific besides the
    split weights, so the switch takes effect on the next forward (re-capture hipGraphs)."""
    global _conv_precision
    if mode not in CONV_PRECISIONS:
        raise ValueError(f"conv precision must be one of {CONV_PRECISIONS}, got {mode!r}")
    prev, _conv_precision = _conv_precision, mode
    return prev


class conv_precision_scope:
    """`with conv_precision_scope(mode):` - the matrix-core mode for the calls inside (None: no change).  What lets two
    HairFast objects of one process run in different modes (`HairFast(args, conv_precision=...)`): the switch itself is
    process-wide, every call of such an object sets it for its own duration and restores it.  Not thread-safe, like the
    reference's single-threaded callers."""

    def __init__(self, mode):
        if mode is not None and mode not in CONV_PRECISIONS:
            raise ValueError(f"conv precision must be one of {CONV_PRECISIONS}, got {mode!r}")
        self.mode, self.prev = mode, None

    def __enter__(self):
        if self.mode is not None:
            self.prev = set_conv_precision(self.mode)
        return self

    def __exit__(self, *exc):
        if self.mode is not None:
            set_conv_precision(self.prev)
        return False


def reference_rng_walk():
    """HAIRFAST_RNG_WALK=reference: consume torch's device RNG in the reference's ORDER - one normal_() per NoiseInjection
    layer (models/stylegan2/model.py:289-291) instead of one per forward, and the FS encoder's discarded generator forward
    (trainer.py:295) run for its 17 draws.  Costs ~16 launches and 148.5 GFLOP per embedded image; the draws still come from
    hipRAND's Philox, so a CUDA run is matched in distribution and call order, not bit for bit.  Read per call."""
    import os

    return os.environ.get("HAIRFAST_RNG_WALK", "") == "reference"


# ---------------------------------------------------------------------------------------
# Batch-invariant plans - THE DEFAULT since round 5 (HAIRFAST_DETERMINISTIC=0 / set_batch_invariant(False) opt out): every
# decision that changes a sample's bits - the K partition (split-K factor) and the kernel family (fp32 split-K / tap-GEMM /
# tiled fp16-core kernels sum in different orders) - is made for a fixed CANONICAL batch (3: the Embedding stage's batch of a
# single swap), never for the batch a sample happens to run in, so that a sample's result has the same bits whatever it is
# batched with: `HairFast.swap_batch` produces exactly the segmentation-mask indices of `HairFast.swap` (north_star: bit-exact
# mask indices; an argmax near a tie otherwise flips with the batch size).  What keeps following the whole launch are the
# things that do NOT change the K order: tile forms, pre-split vs register-staged input, images per GEMM tile - and HOW a K
# partition is executed: a launch that fills the chip by itself walks the slabs inside its blocks (ConvParams::vsplit), a
# batch-1 launch spreads them over the grid and adds them in a second pass, same bits.
# Process-wide, like the library's flag (hf_set_batch_invariant), which `lib()` keeps equal to this setting;
# `HairFast(args, batch_invariant=...)` sets it for the duration of that object's calls.
CANON_BATCH = 3  # conv_common.h kCanonBatch
_batch_invariant = os.environ.get("HAIRFAST_DETERMINISTIC", "1") not in ("", "0")
_lib_flag = False  # what the library currently holds


def batch_invariant():
    return _batch_invariant


def set_batch_invariant(on):
    """Returns the previous setting.  Re-capture hipGraphs after a change (a graph replays the plans it recorded)."""
    global _batch_invariant
    prev, _batch_invariant = _batch_invariant, bool(on)
    lib()  # applies the setting to the library now: plans are made inside C calls that never pass through plan_batch
    return prev


class batch_invariant_scope:
    """`with batch_invariant_scope(on):` - batch-invariant plans on / off for the calls inside (None: no change); what
    `HairFast(args, batch_invariant=...)` wraps its calls in, like conv_precision_scope.  NOT thread-safe either: the flag is
    process-wide (this module's global and the library's g_batch_invariant) - two HairFast objects with different settings
    running in different threads would change each other's plans mid-call and silently lose batch invariance; run such
    objects from one thread (the per-thread split-K counter registration does not change that)."""

    def __init__(self, on):
        self.on, self.prev = on, None

    def __enter__(self):
        if self.on is not None:
            self.prev = set_batch_invariant(self.on)
        return self

    def __exit__(self, *exc):
        if self.on is not None:
            set_batch_invariant(self.prev)
        return False


def plan_batch(batch):
    """The batch count host-side decisions that change a sample's BITS are made with (kernel family: fp32 split-K / tap-GEMM /
    tiled; the library's plan_batch, conv_common.h).  Bit-neutral choices (pre-splitting an input, tile forms) use the real batch."""
    return CANON_BATCH if _batch_invariant else batch
