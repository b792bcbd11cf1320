"""Host-side operators over libb2t_hip.so: thin typed wrappers (torch tensors in, raw pointers out)
and the forward/backward orchestration of the day-layer -> GRU stack -> head -> CTC path.

PyTorch is used for device memory, streams/events and autograd plumbing only; every arithmetic step is a
call into the C ABI (include/b2t.h).  There is no CPU fallback: tensors must live on the HIP device.

Execution plan (model_forward / model_backward).  A GRU layer is a strictly serial chain over time, and
one layer's persistent sweep keeps only H/16 x ceil(B/16) = 128 workgroups busy, each mostly waiting on
the inter-workgroup hand-off.  The time axis is therefore cut into chunks and the layers are software
pipelined over them on per-layer HIP streams: while layer l sweeps chunk c, layer l+1 runs its input
projection GEMM + sweep on chunk c-1, etc.; in the backward pass the weight-gradient GEMMs of layer l run on
that layer's GEMM stream while the layers below are still sweeping.  Dependencies are HIP events; the
caller's stream joins all of them before the function returns.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

import b2t_native as N

ARENA_ALIGN = 1024  # floats; one optimizer chunk (csrc/optimizer.hip)


def _p(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need(t: torch.Tensor, dtype=torch.float32, name="tensor"):
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be on the HIP device (got {t.device}); this package has no CPU path")
    if t.dtype != dtype:
        raise RuntimeError(f"{name} must be {dtype} (got {t.dtype})")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")
    return t


def pad_to(n: int, a: int = ARENA_ALIGN) -> int:
    return (n + a - 1) // a * a


# Live kernel timing for bench.py: when PROFILE["on"], every C-ABI call below is bracketed by HIP
# events recorded on the launch stream (torch's current stream — the stream handed to the library).
PROFILE = {"on": False, "ev": []}


class _Prof:
    def __init__(self, name, flops=0.0, launches=1):
        self.name, self.flops, self.launches = name, flops, launches

    def __enter__(self):
        if PROFILE["on"]:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if PROFILE["on"]:
            self.e1.record()
            PROFILE["ev"].append((self.name, self.flops, self.launches, self.e0, self.e1))
        return False


# ------------------------------------------------------------------------------------------------
# smoothing kernel taps (model_training/data_augmentations.py:19-24), computed on the host
# ------------------------------------------------------------------------------------------------
_TAPS_CACHE: Dict[Tuple[float, int], np.ndarray] = {}


def gauss_taps(std: float, size: int) -> np.ndarray:
    """Impulse response of scipy's gaussian_filter1d (truncate 4 sigma) on a `size`-sample unit
    impulse, entries > 0.01 kept and renormalised — same construction as the reference."""
    key = (float(std), int(size))
    if key not in _TAPS_CACHE:
        r = int(4.0 * float(std) + 0.5)
        x = np.arange(-r, r + 1, dtype=np.float64)
        phi = np.exp(-0.5 / (float(std) ** 2) * x * x)
        phi /= phi.sum()
        resp = np.zeros(int(size), dtype=np.float32)
        c = int(size) // 2
        lo, hi = max(0, c - r), min(int(size), c + r + 1)
        resp[lo:hi] = phi[(lo - (c - r)):(hi - (c - r))].astype(np.float32)
        keep = resp[resp > 0.01]
        _TAPS_CACHE[key] = (keep / np.sum(keep)).astype(np.float32)
    return _TAPS_CACHE[key]


def augment_smooth(x: torch.Tensor, std: float, size: int, padding: str = "same", cut: int = 0,
                   white_std: float = 0.0, offset_std: float = 0.0, seed: int = 0,
                   white_noise: Optional[torch.Tensor] = None, offset_noise: Optional[torch.Tensor] = None,
                   smooth: bool = True) -> torch.Tensor:
    """Fused noise + cut + Gaussian smoothing (rnn_trainer.py:436-484, data_augmentations.py:6-37)."""
    _need(x, name="features")
    B, T, F = x.shape
    taps = gauss_taps(std, size) if smooth else np.ones(1, dtype=np.float32)
    nt = int(taps.shape[0])
    mode = {"same": 0, "valid": 1}[padding]
    Tc = T - cut
    T_out = Tc if mode == 0 else Tc - nt + 1
    if T_out <= 0:
        raise RuntimeError("sequence shorter than the smoothing kernel")
    y = torch.empty((B, T_out, F), dtype=torch.float32, device=x.device)
    tp = (C.c_float * nt)(*[float(v) for v in taps])
    if white_noise is not None:
        _need(white_noise, name="white_noise")
    if offset_noise is not None:
        _need(offset_noise, name="offset_noise")
    with _Prof("augment_smooth"):
        N.check(N.load().b2t_augment_smooth_f32(_p(x), _p(y), B, T, F, int(cut), float(white_std), float(offset_std),
                                                C.c_uint64(seed & (2 ** 64 - 1)), _p(white_noise), _p(offset_noise),
                                                tp, nt, mode, _stream()), "b2t_augment_smooth_f32")
    return y


# ------------------------------------------------------------------------------------------------
# GEMM / reductions
# ------------------------------------------------------------------------------------------------
# Matmul precision: "f32" = exact fp32 MFMA (default; BASELINE config 2), "bf16" = operands rounded to bf16 on the way to
# the matrix cores, fp32 accumulate / output (the reference's `use_amp: true` regime, rnn_args.yaml; B2T_AMP=1 or
# set_amp(True)): all GEMMs and the recurrent products of the persistent sweeps.  Gate math, accumulation, CTC and the
# optimizer stay fp32 either way.
AMP = {"on": os.environ.get("B2T_AMP", "0") not in ("0", "", "false", "False"),
       "sweeps": os.environ.get("B2T_AMP_SWEEPS", "1") != "0"}   # B2T_AMP_SWEEPS=0: bf16 GEMMs only, fp32 recurrent products


def set_amp(on: bool):
    AMP["on"] = bool(on)


def gemm(A, B, Cm, *, M, N_, K, Z=1, a_kc=1, b_kc=1, a_s0=0, a_s1=0, a_div=0, a_sz=0, b_s0=0, b_s1=0, b_div=0,
         b_sz=0, c_s0=0, c_s1=0, c_div=0, c_sz=0, bias=None, bias_sz=0, b_zmap=None, epilogue=0, accumulate=0,
         a_off=0, b_off=0, c_off=0, splitk=1, ws=None, slab="splitk_slab", _splitk=1, _c_ks=0, a_brk=0, a_gap=0):
    """C = A.B^T through b2t_gemm_f32.  splitk>1 (needs `ws`): partial products go to a workspace slab and
    are summed deterministically into C by b2t_colsum_f32 (used for the weight gradients, K = T*B)."""
    if splitk > 1:
        if Z != 1 or epilogue != 0 or c_div != 0 or c_s0 != N_:
            raise RuntimeError("split-K gemm supports Z=1, dense row-major C, no epilogue")
        sl = ws.get(slab, (splitk, M * N_), Cm.device)
        gemm(A, B, sl, M=M, N_=N_, K=K, a_kc=a_kc, b_kc=b_kc, a_s0=a_s0, a_s1=a_s1, a_div=a_div, b_s0=b_s0,
             b_s1=b_s1, b_div=b_div, c_s0=N_, bias=bias, a_off=a_off, b_off=b_off, _splitk=splitk, _c_ks=M * N_,
             a_brk=a_brk, a_gap=a_gap)
        N.check(N.load().b2t_slab_reduce_f32(_p(sl), splitk, M * N_, C.c_void_p(Cm.data_ptr() + 4 * c_off), accumulate,
                                             _stream()), "b2t_slab_reduce_f32")
        return
    d = N.GemmDesc()
    d.A = A.data_ptr() + 4 * a_off
    d.B = B.data_ptr() + 4 * b_off
    d.C = Cm.data_ptr() + 4 * c_off
    d.bias = bias.data_ptr() if bias is not None else None
    d.M, d.N, d.K, d.Z = M, N_, K, Z
    d.a_kcontig, d.b_kcontig = a_kc, b_kc
    d.a_s0, d.a_s1, d.a_div, d.a_sz = a_s0, a_s1, a_div, a_sz
    d.b_s0, d.b_s1, d.b_div, d.b_sz = b_s0, b_s1, b_div, b_sz
    d.c_s0, d.c_s1, d.c_div, d.c_sz = c_s0, c_s1, c_div, c_sz
    d.b_zmap = b_zmap.data_ptr() if b_zmap is not None else None
    d.bias_sz = bias_sz
    d.epilogue, d.accumulate = epilogue, accumulate
    d.splitk, d.c_ks = _splitk, _c_ks
    d.a_brk, d.a_gap = a_brk, a_gap
    if AMP["on"]:
        with _Prof(f"gemm_bf16_kernel<{int(bool(a_kc))},{int(bool(b_kc))}>", 2.0 * M * N_ * K * Z):
            N.check(N.load().b2t_gemm_bf16_f32(C.byref(d), _stream()), "b2t_gemm_bf16_f32")
        return
    with _Prof(f"gemm_f32_kernel<{int(bool(a_kc))},{int(bool(b_kc))}>", 2.0 * M * N_ * K * Z):   # one name per rocprof symbol
        N.check(N.load().b2t_gemm_f32(C.byref(d), _stream()), "b2t_gemm_f32")


def splitk_for(M: int, Nn: int, K: int, target_blocks: int = 1024) -> int:
    """Number of K slices so that a weight-gradient GEMM (few 128x128 output tiles, very long K) fills the
    chip: ~4 workgroups per CU on 256 CUs, each slice at least 256 deep."""
    tiles = ((M + 127) // 128) * ((Nn + 127) // 128)
    return int(max(1, min(target_blocks // max(1, tiles), K // 256)))


def colsum(x, rows, cols, ld, out, accumulate=0, Z=1, x_sz=0, out_sz=0, x_off=0, out_off=0):
    lib = N.load()
    nbytes = lib.b2t_colsum_ws_bytes(rows, cols) * Z
    ws = torch.empty((nbytes // 4 + 1,), dtype=torch.float32, device=x.device)
    N.check(lib.b2t_colsum_f32(C.c_void_p(x.data_ptr() + 4 * x_off), rows, cols, ld,
                               C.c_void_p(out.data_ptr() + 4 * out_off), accumulate, _p(ws), Z, x_sz, out_sz,
                               _stream()), "b2t_colsum_f32")


def dropout(x, y, n, p, seed, elem0=0, x_off=0, y_off=0):
    N.check(N.load().b2t_dropout_f32(C.c_void_p(x.data_ptr() + 4 * x_off), C.c_void_p(y.data_ptr() + 4 * y_off), n,
                                     float(p), C.c_uint64(seed), elem0, _stream()), "b2t_dropout_f32")


# ------------------------------------------------------------------------------------------------
# model description, parameter views, workspace
# ------------------------------------------------------------------------------------------------
class ModelDims:
    def __init__(self, neural_dim, n_units, n_days, n_classes, n_layers, patch_size, patch_stride):
        self.F, self.H, self.D, self.C, self.L = neural_dim, n_units, n_days, n_classes, n_layers
        self.patch, self.stride = patch_size, patch_stride
        self.In0 = neural_dim * patch_size if patch_size > 0 else neural_dim
        if self.H % 16 != 0:
            raise RuntimeError(f"n_units={self.H} must be a multiple of 16 for the MFMA recurrent tiles")
        if self.F % 4 != 0:
            raise RuntimeError(f"neural_dim={self.F} must be a multiple of 4")

    def out_T(self, T):
        return (T - self.patch) // self.stride + 1 if self.patch > 0 else T


class Params:
    """Views of the model's parameter arena needed by the kernels."""
    def __init__(self, day_w, day_b, day_w_stride, day_b_stride, w_ih, w_hh, b_ih, b_hh, out_w, out_b, h0):
        self.day_w, self.day_b = day_w, day_b                  # arena views starting at day 0
        self.day_w_stride, self.day_b_stride = day_w_stride, day_b_stride
        self.w_ih, self.w_hh, self.b_ih, self.b_hh = w_ih, w_hh, b_ih, b_hh
        self.out_w, self.out_b, self.h0 = out_w, out_b, h0


class Grads(Params):
    """Destination views for parameter gradients (normally views of the model's gradient arena)."""


# GRU sweep mode: 0 = one launch per time step; 1 = persistent single-launch sweep with counter hand-off
# (csrc/gru_persistent.hip); 2 = persistent sweep with data-tagged granule hand-off (csrc/gru_granule.hip).
# -1 = choose per call: mode 1 whenever its (H/16) x ceil(B/16) workgroups can be co-resident, else 0.
# (Mode 2 is parity-clean but measured SLOWER than mode 1 on MI355X — 5.6 vs 4.6 us/step at H=512, B=64: with a
# 32 KB payload per workgroup-step every retry of the tagged read is a full fabric round trip; it stays opt-in.)
GRU_MODE = {"value": int(os.environ.get("B2T_GRU_MODE", "-1"))}
MAX_RESIDENT_WGS = 256   # MI355X: 256 CUs; a persistent sweep needs all its workgroups resident at once
# Number of time chunks the layers are software-pipelined over (1 = layer-by-layer, no side streams).
PIPELINE = {"chunks": int(os.environ.get("B2T_CHUNKS", "6")), "min_chunk": 16,
            "sweep_streams": int(os.environ.get("B2T_SWEEP_STREAMS", "64")),
            "bwd_sweeps": int(os.environ.get("B2T_BWD_SWEEPS", "64")),
            "sweep_priority": int(os.environ.get("B2T_SWEEP_PRIORITY", "0")),
            # sub-chunk flags (SweepFlags, csrc/gru_sync.h): steps per sub-chunk, 0 = event-per-chunk hand-over
            "sub": int(os.environ.get("B2T_SUB", "0")),
            # experiment: run the weight-gradient GEMMs of layers >= 1 after the last backward sweep instead of under the sweeps
            "defer_wgrad": int(os.environ.get("B2T_DEFER_WGRAD", "0"))}


def bwd_mode_for(fwd_mode: int) -> int:
    """Backward sweep mode that goes with a forward mode (B2T_GRU_BWD_MODE overrides): the granule forward (2) pairs
    with the counter backward (1); the pipelined forward (3) with the pipelined backward (3)."""
    e = os.environ.get("B2T_GRU_BWD_MODE")
    if e is not None and fwd_mode >= 1:
        return int(e)
    return {0: 0, 1: 1, 2: 1, 3: 3}[fwd_mode]


def gru_mode_for(B: int, H: int) -> int:
    m = GRU_MODE["value"]
    if m in (0, 1, 2, 3):
        return m
    if m == 4:   # fused stack where the shape is covered (b2t_gru_stack_*), per-layer persistent sweeps otherwise
        return 1
    return 1 if (H // 16) * ((B + 15) // 16) <= MAX_RESIDENT_WGS and H <= 1024 else 0


def stack_wanted() -> bool:
    """B2T_GRU_MODE=4: run the GRU stack as one persistent launch per direction (csrc/gru_stack.hip)."""
    return GRU_MODE["value"] == 4


def gru_stack_forward(dims, prm, gi0, outs, outs_d, reserves, Tp, B, rnn_drop, seed, ws, dev) -> bool:
    """All layers' recurrences (and the input projections of layers >= 1) in one launch on the current stream.
    outs[l][0] must hold the initial state.  Returns False (nothing launched) when the shape is not covered."""
    H, L = dims.H, dims.L
    d = N.GruStackDesc()
    d.T, d.B, d.H, d.L = Tp, B, H, L
    d.gi0 = gi0.data_ptr()
    d.bf16 = 1 if (AMP["on"] and AMP.get("sweeps", True)) else 0
    for l in range(L):
        d.w_hh[l] = prm.w_hh[l].data_ptr(); d.b_hh[l] = prm.b_hh[l].data_ptr()
        d.w_ih[l] = prm.w_ih[l].data_ptr(); d.b_ih[l] = prm.b_ih[l].data_ptr()
        d.out[l] = outs[l].data_ptr()
        d.reserve[l] = reserves[l].data_ptr() if reserves[l] is not None else None
        if outs_d[l] is not outs[l]:
            # nn.GRU inter-layer dropout (rnn_model.py:70): the factors are made here (whole chip, one write pass), the
            # producing layer multiplies its tile by them -- Philox inside the sweep was its longest VALU chain
            mask = ws.get(f"stack_mask{l}", (Tp, B, H), dev)
            N.check(N.load().b2t_dropout_mask_f32(_p(mask), Tp * B * H, float(rnn_drop),
                                                  C.c_uint64((seed * 1000003 + 101 + l) & 0xFFFFFFFFFFFFFFFF), 0, _stream()),
                    "b2t_dropout_mask_f32")
            d.out_drop[l] = outs_d[l].data_ptr()
            d.drop_mask[l] = mask.data_ptr()
    with _Prof("gru_stack_fwd", 2.0 * Tp * B * 3 * H * H * (2 * L - 1), 1):
        rc = N.load().b2t_gru_stack_fwd_f32(C.byref(d), _p(ws.sync_ws(0, Tp, dev, B, H, "stk")), _stream())
    if rc == 4:
        return False
    N.check(rc, "b2t_gru_stack_fwd_f32")
    return True


GRU_BF16 = 0x100   # B2T_GRU_BF16 (include/b2t.h): bf16 operands of the recurrent product, persistent mode 1


GRU_WIDE = 0x200   # B2T_GRU_WIDE: 32 hidden units per workgroup (bf16 operands only)
# which sweeps run with 32-unit workgroups under AMP: "" none, "f" forward, "b" backward, "fb" both (B2T_AMP_WIDE;
# measured at C2: 18.4 / 17.5 / 17.1 / 16.1 ms per step)
AMP["wide"] = os.environ.get("B2T_AMP_WIDE", "fb")


def sweep_mode_arg(mode: int, H: int = 0, direction: str = "f") -> int:
    """`mode` argument of b2t_gru_layer_fwd/bwd_f32: under set_amp(True) the persistent sweeps take bf16 operands."""
    if not (AMP["on"] and AMP.get("sweeps", True) and mode == 1):
        return mode
    wide = GRU_WIDE if (direction in AMP["wide"] and H % 32 == 0 and H <= 512) else 0   # (LDS staging of H > 512 exceeds 64 KB)
    return mode | GRU_BF16 | wide


def gru_sync_check(sync_ws, T: int, B: int):
    """Raise if the last persistent sweep on sync_ws reported a hand-off timeout (synchronises)."""
    st = C.c_int(0)
    N.check(N.load().b2t_gru_sync_status(_p(sync_ws), T, B, C.byref(st), _stream()), "b2t_gru_sync_status")
    if st.value != 0:
        raise RuntimeError("persistent GRU sweep: inter-workgroup hand-off timed out (results invalid)")


def gru_sync_check_all(ws, L: int, Tp: int, B: int, device, H: int = 512):
    for l in range(L):
        gru_sync_check(ws.sync_ws(l, Tp, device, B, H), Tp, B)
        gru_sync_check(ws.sync_ws(l, Tp, device, B, H, "b"), Tp, B)
    gru_sync_check(ws.sync_ws(0, Tp, device, B, H, "stk"), Tp, B)


class Workspace:
    """Shape-keyed scratch buffers (allocated once per shape from torch's caching allocator) plus the
    per-layer side streams of the pipelined execution plan."""
    def __init__(self):
        self.bufs: Dict[Tuple, torch.Tensor] = {}
        self.streams: Dict[Tuple, List[torch.cuda.Stream]] = {}

    def get(self, name, shape, device, dtype=torch.float32):
        key = (name, tuple(shape), str(device), dtype)
        t = self.bufs.get(key)
        if t is None:
            t = (torch.zeros if dtype == torch.int32 else torch.empty)(shape, dtype=dtype, device=device)
            self.bufs[key] = t
        return t

    def layer_streams(self, L, device):
        key = (L, str(device))
        if key not in self.streams:
            # One sweep stream per layer by default.  B2T_SWEEP_STREAMS=2 (with B2T_SWEEP_EXCLUSIVE=1: one persistent
            # workgroup per CU) bounds the sweeps in flight to two — measured slower inside the full step, see DESIGN.md.
            nsw = max(1, min(L, PIPELINE["sweep_streams"]))
            base = [torch.cuda.Stream(device=device, priority=(-1 if PIPELINE["sweep_priority"] else 0)) for _ in range(nsw)]
            self.streams[key] = ([base[l % nsw] for l in range(L)],
                                 [torch.cuda.Stream(device=device) for _ in range(L)])
        return self.streams[key]

    def check_sync(self):
        """Raise if any persistent sweep that used this workspace reported a hand-off timeout (synchronises)."""
        for key, buf in self.bufs.items():
            if str(key[0]).startswith("gru_sync") and int(buf[0].item()) != 0:
                raise RuntimeError(f"persistent GRU sweep ({key[0]}): inter-workgroup hand-off timed out — results invalid")

    def wgrad_streams(self, L, device):
        # The weight-gradient GEMMs share the per-layer GEMM streams: the plan must not use more HIP streams than the
        # runtime has hardware queues (GPU_MAX_HW_QUEUES=16 is set by bench.py / the trainer).  Streams that share a
        # hardware queue serialise, and a persistent sweep stuck behind the kernel it waits for costs milliseconds
        # (measured: 16 streams -> 86 ms/step instead of 31).  2L+1 = 11 streams at L=5.
        return self.layer_streams(L, device)[1]

    def sync_ws(self, l, Tp, device, B=64, H=512, tag=""):
        return self.get(f"gru_sync{tag}{l}", (N.load().b2t_gru_ws_bytes(Tp, B, H) // 4 + 16,), device, torch.int32)


def sub_ranges(n: int, sub: int, from_end: bool = False) -> List[Tuple[int, int]]:
    """Sub-chunks of a launch of n steps, in the order the sweep visits them: forward from step 0, backward from step
    n-1 (the k-th range then is [n-(k+1)*sub, n-k*sub))."""
    if from_end:
        return [(max(0, n - (k + 1) * sub), n - k * sub) for k in range((n + sub - 1) // sub)]
    return [(k * sub, min(n, (k + 1) * sub)) for k in range((n + sub - 1) // sub)]


def stream_write_value(t: torch.Tensor, index: int, value: int):
    N.check(N.load().b2t_stream_write_value32(C.c_void_p(t.data_ptr() + 4 * index), value, _stream()), "b2t_stream_write_value32")


def stream_wait_value(t: torch.Tensor, index: int, value: int):
    N.check(N.load().b2t_stream_wait_value32_gte(C.c_void_p(t.data_ptr() + 4 * index), value, _stream()),
            "b2t_stream_wait_value32_gte")


def time_chunks(Tp: int, B: int = 64, H: int = 512) -> List[Tuple[int, int]]:
    """Time chunks of the layer pipeline.  Pipelining pays only when the sweeps of two layers can be resident together:
    a sweep is (H/16) * ceil(B/16) workgroups that must all run at once, and the chip holds 2 backward-sweep workgroups
    per CU up to H = 512 but only 1 beyond (the W_hh slice takes the whole register file).  At H = 768, B = 64 (192
    workgroups of 256 slots) concurrent sweeps just block each other: 37-93 ms per step with 6 chunks against 19.8 ms
    with the layers in sequence (tools/bench_c3.py)."""
    wgs = (H // 16) * ((B + 15) // 16)
    slots = MAX_RESIDENT_WGS * (2 if H <= 512 else 1)
    if 2 * wgs > slots and "B2T_CHUNKS" not in os.environ:
        return [(0, Tp)]
    n = max(1, min(PIPELINE["chunks"], Tp // max(1, PIPELINE["min_chunk"])))
    ch = (Tp + n - 1) // n
    return [(t0, min(Tp, t0 + ch)) for t0 in range(0, Tp, ch)]


class ForwardCtx:
    pass


def _ev(stream=None):
    e = torch.cuda.Event()
    e.record(stream if stream is not None else torch.cuda.current_stream())
    return e


# ------------------------------------------------------------------------------------------------
# forward
# ------------------------------------------------------------------------------------------------
def model_forward(dims: ModelDims, prm: Params, x: torch.Tensor, day_idx: torch.Tensor,
                  states: Optional[torch.Tensor], ws: Workspace, save: bool,
                  in_drop: float = 0.0, rnn_drop: float = 0.0, seed: int = 0, reuse_saved: bool = False):
    """day layer -> (patch) -> L x GRU -> head.  x [B,T,F] fp32 on device, day_idx int32 [B].
    Returns logits [B,T',C], hidden [L,B,H] and (if save) the context for model_backward.
    Mirrors GRUDecoder.forward (model_training/rnn_model.py:88-134)."""
    lib = N.load()
    _need(x, name="x")
    _need(day_idx, torch.int32, "day_idx")
    B, T, F = x.shape
    H, L, Cc = dims.H, dims.L, dims.C
    if F != dims.F:
        raise RuntimeError(f"input feature dim {F} != neural_dim {dims.F}")
    Tp = dims.out_T(T)
    if Tp <= 0:
        raise RuntimeError("sequence shorter than patch_size")
    dev = x.device
    main = torch.cuda.current_stream()

    def sbuf(name, shape):
        # saved-for-backward buffers: workspace-owned only when the caller guarantees one
        # forward/backward in flight (the trainer's fused step); fresh allocations otherwise.
        if save and reuse_saved:
            return ws.get(name, shape, dev)
        return torch.empty(shape, dtype=torch.float32, device=dev)

    # 1. day layer: U[b] = softsign(x[b] @ W[day[b]] + c[day[b]])   (rnn_model.py:95-99)
    U = sbuf("U", (B, T, F))
    gemm(x, prm.day_w, U, M=T, N_=F, K=F, Z=B, a_kc=1, a_s0=F, a_sz=T * F, b_kc=0, b_s0=F, b_sz=prm.day_w_stride,
         c_s0=F, c_sz=T * F, bias=prm.day_b, bias_sz=prm.day_b_stride, b_zmap=day_idx, epilogue=1)
    Ud = U
    if in_drop > 0:
        Ud = sbuf("Ud", (B, T, F))
        dropout(U, Ud, U.numel(), in_drop, seed * 1000003 + 17)

    mode = gru_mode_for(B, H)
    chunks = time_chunks(Tp, B, H)
    s_sweep, s_gemm = ws.layer_streams(L, dev)
    piped = len(chunks) > 1
    outs = [sbuf(f"out{l}", (Tp + 1, B, H)) for l in range(L)]
    outs_d = [sbuf(f"outd{l}", (Tp + 1, B, H)) if (rnn_drop > 0 and l < L - 1) else outs[l] for l in range(L)]
    reserves = [sbuf(f"res{l}", (Tp, B, 4 * H)) if save else None for l in range(L)]
    gis = [ws.get(f"gi{l if piped else 0}", (Tp, B, 3 * H), dev) for l in range(L)]
    hidden = torch.empty((L, B, H), dtype=torch.float32, device=dev)
    a_s0_l0 = dims.stride * F if dims.patch > 0 else F
    stacked = False
    if stack_wanted() and mode == 1:
        # mode 4: layer 0's projection for the whole sequence, then ONE launch for the L recurrences (layers >= 1
        # project inside the sweep, dropout is applied by the producing layer)
        for l in range(L):
            outs[l][0].copy_(prm.h0.view(1, H).expand(B, H) if states is None else states[l])
        gemm(Ud, prm.w_ih[0], gis[0], M=Tp, N_=3 * H, K=dims.In0, Z=B, a_kc=1, a_s0=a_s0_l0, a_sz=T * F,
             b_kc=1, b_s0=dims.In0, c_s0=B * 3 * H, c_sz=3 * H, bias=prm.b_ih[0])
        stacked = gru_stack_forward(dims, prm, gis[0], outs, outs_d, reserves, Tp, B, rnn_drop, seed, ws, dev)
        if stacked:
            for l in range(L):
                hidden[l].copy_(outs[l][Tp])
    ev0 = _ev(main)
    if piped:
        for s in s_sweep + s_gemm:
            s.wait_event(ev0)
    for l in range(L if not stacked else 0):   # slot 0 = initial state, so outs[l][0:T'] is the h_{t-1} matrix
        # (on the layer's sweep stream: five small broadcast copies in front of the first GEMM were ~0.25 ms of step)
        with torch.cuda.stream(s_sweep[l] if piped else main):
            if states is None:
                outs[l][0].copy_(prm.h0.view(1, H).expand(B, H))
            else:
                outs[l][0].copy_(states[l])
    ev_sw: List[List[Optional[torch.cuda.Event]]] = [[None] * len(chunks) for _ in range(L)]
    # cells (chunk c, layer l) are enqueued diagonal by diagonal (c + l), a topological order in which the two
    # sweep streams never wait on work that is queued behind them
    # Sub-chunk flags: the consumer layer trails its producer by SUB steps instead of a whole chunk launch (the
    # projection GEMM of a chunk no longer sits between two dependent sweeps).  Flag words only ever grow (epoch).
    SUB = PIPELINE["sub"]
    flagged = piped and SUB > 0 and mode == 1 and rnn_drop == 0
    if flagged:
        nsub_max = max((t1 - t0 + SUB - 1) // SUB for t0, t1 in chunks)
        fflags = ws.get("fwd_flags", (2, L, len(chunks), nsub_max), dev, torch.int32)   # [ready|done][l][c][k]
        ws.epoch = getattr(ws, "epoch", 0) + 1
        epoch = ws.epoch
        fidx = lambda kind, l, c, k: ((kind * L + l) * len(chunks) + c) * nsub_max + k
    for c, l in sorted(((c, l) for c in range(len(chunks)) for l in range(L if not stacked else 0)),
                       key=lambda cl: (cl[0] + cl[1], cl[1])):
        t0, t1 = chunks[c]
        n = t1 - t0
        if flagged:
            sg, ss = s_gemm[l], s_sweep[l]
            with torch.cuda.stream(sg):
                for k, (s0, s1) in enumerate(sub_ranges(n, SUB)):
                    if l == 0:
                        gemm(Ud, prm.w_ih[0], gis[0], M=s1 - s0, N_=3 * H, K=dims.In0, Z=B, a_kc=1, a_s0=a_s0_l0, a_sz=T * F,
                             a_off=(t0 + s0) * a_s0_l0, b_kc=1, b_s0=dims.In0, c_s0=B * 3 * H, c_sz=3 * H,
                             c_off=(t0 + s0) * B * 3 * H, bias=prm.b_ih[0])
                    else:
                        stream_wait_value(fflags, fidx(1, l - 1, c, k), epoch)
                        gemm(outs[l - 1], prm.w_ih[l], gis[l], M=(s1 - s0) * B, N_=3 * H, K=H, a_kc=1, a_s0=H,
                             a_off=(1 + t0 + s0) * B * H, b_kc=1, b_s0=H, c_s0=3 * H, c_off=(t0 + s0) * B * 3 * H,
                             bias=prm.b_ih[l])
                    stream_write_value(fflags, fidx(0, l, c, k), epoch)
            with torch.cuda.stream(ss):
                if l > 0:   # launch only once the producer sweep is up and running (residency: DESIGN.md 4b)
                    stream_wait_value(fflags, fidx(1, l - 1, c, 0), epoch)
                res_ptr = C.c_void_p(reserves[l].data_ptr() + 4 * t0 * B * 4 * H) if save else None
                with _Prof("gru_sweep_fwd", 2.0 * n * B * 3 * H * H, 1):
                    N.check(lib.b2t_gru_layer_fwd_flagged_f32(
                        C.c_void_p(gis[l].data_ptr() + 4 * t0 * B * 3 * H), _p(prm.w_hh[l]), _p(prm.b_hh[l]),
                        C.c_void_p(outs[l].data_ptr() + 4 * t0 * B * H),
                        C.c_void_p(outs[l].data_ptr() + 4 * (1 + t0) * B * H), res_ptr,
                        _p(hidden[l]) if t1 == Tp else None, n, B, H, _p(ws.sync_ws(l, Tp, dev, B, H)),
                        C.c_void_p(fflags.data_ptr() + 4 * fidx(0, l, c, 0)),
                        C.c_void_p(fflags.data_ptr() + 4 * fidx(1, l, c, 0)) if l < L - 1 else None,
                        SUB, epoch, _stream()), "b2t_gru_layer_fwd_flagged_f32")
                ev_sw[l][c] = _ev(ss)
            continue
        sg = s_gemm[l] if piped else main
        ss = s_sweep[l] if piped else main
        # 2. input projection gi = in_t W_ih^T + b_ih for this chunk, time-major [T'][B][3H]
        with torch.cuda.stream(sg):
            if l == 0 and n * B <= 512 and dims.In0 >= 2048:
                # streaming-sized calls (a few frames, patch input K = 7168): one GEMM over all (t, b) rows through
                # the two-level row map, K split over the chip -- as B per-sentence GEMMs of M = n rows the K loop
                # runs serially in 18 workgroups per sentence (0.4 ms of a 2 ms step)
                gemm(Ud, prm.w_ih[0], gis[0], M=n * B, N_=3 * H, K=dims.In0, a_kc=1, a_div=B, a_s1=a_s0_l0, a_s0=T * F,
                     a_off=t0 * a_s0_l0, b_kc=1, b_s0=dims.In0, c_s0=3 * H, c_off=t0 * B * 3 * H, bias=prm.b_ih[0],
                     splitk=max(1, min(16, dims.In0 // 448)), ws=ws, slab="splitk_slab_gi0")
            elif l == 0:
                gemm(Ud, prm.w_ih[0], gis[0], M=n, N_=3 * H, K=dims.In0, Z=B, a_kc=1, a_s0=a_s0_l0, a_sz=T * F,
                     a_off=t0 * a_s0_l0, b_kc=1, b_s0=dims.In0, c_s0=B * 3 * H, c_sz=3 * H, c_off=t0 * B * 3 * H,
                     bias=prm.b_ih[0])
            else:
                if piped:
                    sg.wait_event(ev_sw[l - 1][c])
                src = outs[l - 1]
                if outs_d[l - 1] is not outs[l - 1]:   # nn.GRU inter-layer dropout (rnn_model.py:70)
                    dropout(outs[l - 1], outs_d[l - 1], n * B * H, rnn_drop, seed * 1000003 + 101 + (l - 1),
                            elem0=t0 * B * H, x_off=(1 + t0) * B * H, y_off=(1 + t0) * B * H)
                    src = outs_d[l - 1]
                small = n * B <= 512 and H >= 384      # streaming-sized call: split K (one 128-row tile otherwise)
                gemm(src, prm.w_ih[l], gis[l], M=n * B, N_=3 * H, K=H, a_kc=1, a_s0=H, a_off=(1 + t0) * B * H,
                     b_kc=1, b_s0=H, c_s0=3 * H, c_off=t0 * B * 3 * H, bias=prm.b_ih[l],
                     **(dict(splitk=max(1, H // 192), ws=ws, slab="splitk_slab_gi") if small else {}))
            ev_gi = _ev(sg) if piped else None
        # 3. recurrent sweep over the chunk, continuing from outs[l][t0] = h_{t0-1}
        with torch.cuda.stream(ss):
            if piped:
                ss.wait_event(ev_gi)
            res_ptr = C.c_void_p(reserves[l].data_ptr() + 4 * t0 * B * 4 * H) if save else None
            with _Prof("gru_sweep_fwd", 2.0 * n * B * 3 * H * H, n if mode == 0 else 1):
                N.check(lib.b2t_gru_layer_fwd_f32(
                    C.c_void_p(gis[l].data_ptr() + 4 * t0 * B * 3 * H), _p(prm.w_hh[l]), _p(prm.b_hh[l]),
                    C.c_void_p(outs[l].data_ptr() + 4 * t0 * B * H),
                    C.c_void_p(outs[l].data_ptr() + 4 * (1 + t0) * B * H), res_ptr,
                    _p(hidden[l]) if t1 == Tp else None, n, B, H, sweep_mode_arg(mode, H, "f"),
                    _p(ws.sync_ws(l, Tp, dev, B, H)) if mode >= 1 else None, _stream()), "b2t_gru_layer_fwd_f32")
            if piped:
                ev_sw[l][c] = _ev(ss)
    if piped and not stacked:
        # One join is enough: the last chunk of the top layer's sweep transitively depends on every GEMM and sweep
        # enqueued above.  (Each wait is a barrier packet the command processor works through one by one: the 10-15
        # joins that used to sit here and at the end of the backward pass cost ~0.3 ms of idle chip each.)
        main.wait_event(ev_sw[L - 1][-1])

    # 4. head: logits[b,t,:] = out W^T + b  (rnn_model.py:129), written batch-first
    logits = torch.empty((B, Tp, Cc), dtype=torch.float32, device=dev)
    gemm(outs[L - 1], prm.out_w, logits, M=Tp * B, N_=Cc, K=H, a_kc=1, a_s0=H, a_off=B * H, b_kc=1, b_s0=H,
         c_div=B, c_s1=Cc, c_s0=Tp * Cc, bias=prm.out_b)
    if not save:
        return logits, hidden, None
    ctx = ForwardCtx()
    ctx.x, ctx.day_idx, ctx.U, ctx.Ud = x, day_idx, U, Ud
    ctx.outs, ctx.outs_d, ctx.reserves = outs, outs_d, reserves
    ctx.B, ctx.T, ctx.Tp = B, T, Tp
    ctx.in_drop, ctx.rnn_drop, ctx.seed = in_drop, rnn_drop, seed
    ctx.custom_states = states is not None
    ctx.chunks, ctx.mode = chunks, mode
    return logits, hidden, ctx


# ------------------------------------------------------------------------------------------------
# backward
# ------------------------------------------------------------------------------------------------
def model_backward(dims: ModelDims, prm: Params, grd: Grads, ctx: ForwardCtx, dlogits: torch.Tensor, ldd: int,
                   ws: Workspace, dhidden: Optional[torch.Tensor] = None, want_dstates: bool = False,
                   bucket_cb=None):
    """Gradients of every parameter given dlogits [B,T',ldd] (ldd multiple of 4, >= C).
    Overwrites the destinations in `grd` (days absent from the batch are not touched).
    Follows SURVEY Appendix A2/A3; replaces loss.backward() at rnn_trainer.py:547.
    bucket_cb(name) is called (on the stream that produced them) as soon as a gradient bucket is complete:
    "head", "layer{l}", "h0", "day" — the data-parallel reducer hooks its all-reduce there."""
    lib = N.load()
    B, T, Tp = ctx.B, ctx.T, ctx.Tp
    F, H, L, Cc = dims.F, dims.H, dims.L, dims.C
    dev = dlogits.device
    M = Tp * B
    main = torch.cuda.current_stream()
    chunks, mode = ctx.chunks, ctx.mode
    piped = len(chunks) > 1
    s_sweep, s_gemm = ws.layer_streams(L, dev)
    # Residency: a backward sweep workgroup needs ~250 VGPRs/lane -> 2 per CU -> 512 slots on the chip, and L concurrent
    # sweeps (5 x 128 workgroups) do not all fit.  That is safe: the 16-row groups of a sweep are independent
    # recurrences and workgroups are dispatched in grid order, so a partially resident sweep still has every row group
    # but its last one complete; complete groups finish and free their slots, and the at most one incomplete group per
    # sweep holds < H/16 slots (4 per XCD at H=512, of 64).  PIPELINE["bwd_sweeps"] (default: one per layer) can bound
    # the sweeps in flight by sharing streams.  (The hand-off timeouts once seen with 5 in flight were a parity-flip
    # race in the kernel epilogue, fixed in gru_persistent.hip:finish_call, not a residency deadlock.)
    nbs = max(1, min(L, PIPELINE["bwd_sweeps"]))
    s_sweep = [s_sweep[l % nbs] for l in range(L)]
    s_wg = ws.wgrad_streams(L, dev)
    nc = len(chunks)

    # head: d_out[t,b,:] = dlogits[b,t,:] W_out ; dW_out = dlogits^T out ; db_out = colsum
    dYs = [ws.get(f"dY{l}", (Tp, B, H), dev) for l in range(L)]
    gemm(dlogits, prm.out_w, dYs[L - 1], M=M, N_=H, K=Cc, a_kc=1, a_div=B, a_s1=ldd, a_s0=Tp * ldd, b_kc=0, b_s0=H,
         c_s0=H)
    ev_top = _ev(main)
    gemm(dlogits, ctx.outs[L - 1], grd.out_w, M=Cc, N_=H, K=M, a_kc=0, a_div=B, a_s1=ldd, a_s0=Tp * ldd, b_kc=0, b_s0=H,
         b_off=B * H, c_s0=H, splitk=splitk_for(Cc, H, M), ws=ws, slab="splitk_slab_head")
    colsum(dlogits, B * Tp, Cc, ldd, grd.out_b)
    if bucket_cb:
        bucket_cb("head")

    dGs = [ws.get(f"dG{l if piped else 0}", (Tp, B, 4 * H), dev) for l in range(L)]
    dh_init = ws.get("dh_init", (L, B, H), dev)
    carries = [ws.get(f"carry{l}", (2, B, H), dev) for l in range(L)]
    scratch = [ws.get(f"bwd_scratch{l}", (B, H), dev) for l in range(L)]
    whh_ts = [ws.get(f"whh_t{l if piped else 0}", (H, 3 * H), dev) for l in range(L)]
    dU = ws.get("dU", (B, T, F), dev)
    dV = ws.get("dV", (B, Tp, dims.In0), dev) if dims.patch > 0 else None
    ev_dx: List[List[Optional[torch.cuda.Event]]] = [[None] * nc for _ in range(L)]
    ev_bs: List[List[Optional[torch.cuda.Event]]] = [[None] * nc for _ in range(L)]
    ev_wt = [None] * L
    if piped:
        # W_hh^T for the backward sweeps depends on the parameters only: enqueued BEFORE the streams wait for the head
        # (they run while the CTC kernel has the chip to itself instead of in front of the first backward sweep)
        for l in range(L):
            with torch.cuda.stream(s_gemm[l]):
                N.check(lib.b2t_transpose_f32(_p(prm.w_hh[l]), _p(whh_ts[l]), 3 * H, H, _stream()), "b2t_transpose_f32")
                ev_wt[l] = _ev(s_gemm[l])
        for s in s_sweep + s_gemm:
            s.wait_event(ev_top)

    def dx_gemm(l, t0, n):
        """dIn = dGi W_ih for rows of chunk [t0,t0+n): into dY[l-1] (l>0) or dU / dV (l == 0)."""
        a_off = t0 * B * 4 * H
        # dGi = dG[:, 0:2H] ++ dG[:, 3H:4H] is one A operand with a gap (a_brk / a_gap), K = 3H
        gap = dict(a_brk=2 * H, a_gap=H) if (2 * H) % 16 == 0 else None
        if l > 0 and gap:
            gemm(dGs[l], prm.w_ih[l], dYs[l - 1], M=n * B, N_=H, K=3 * H, a_kc=1, a_s0=4 * H, b_kc=0, b_s0=H, c_s0=H,
                 c_off=t0 * B * H, a_off=a_off, **gap)
        elif l > 0:
            kw = dict(M=n * B, N_=H, a_kc=1, a_s0=4 * H, b_kc=0, b_s0=H, c_s0=H, c_off=t0 * B * H)
            gemm(dGs[l], prm.w_ih[l], dYs[l - 1], K=2 * H, a_off=a_off, **kw)
            gemm(dGs[l], prm.w_ih[l], dYs[l - 1], K=H, a_off=a_off + 3 * H, b_off=2 * H * H, accumulate=1, **kw)
        elif gap:
            In = dims.In0
            if dims.patch > 0:
                dst, kw = dV, dict(c_div=B, c_s1=In, c_s0=Tp * In, c_off=t0 * In)
            else:
                dst, kw = dU, dict(c_div=B, c_s1=F, c_s0=T * F, c_off=t0 * F)
            gemm(dGs[0], prm.w_ih[0], dst, M=n * B, N_=In, K=3 * H, a_kc=1, a_s0=4 * H, a_off=a_off, b_kc=0, b_s0=In, **gap, **kw)
        else:
            In = dims.In0
            if dims.patch > 0:
                dst, kw = dV, dict(c_div=B, c_s1=In, c_s0=Tp * In, c_off=t0 * In)
            else:
                dst, kw = dU, dict(c_div=B, c_s1=F, c_s0=T * F, c_off=t0 * F)
            gemm(dGs[0], prm.w_ih[0], dst, M=n * B, N_=In, K=2 * H, a_kc=1, a_s0=4 * H, a_off=a_off, b_kc=0, b_s0=In, **kw)
            gemm(dGs[0], prm.w_ih[0], dst, M=n * B, N_=In, K=H, a_kc=1, a_s0=4 * H, a_off=a_off + 3 * H, b_kc=0, b_s0=In,
                 b_off=2 * H * In, accumulate=1, **kw)

    deferred = []
    for c, l in sorted(((c, l) for c in range(nc) for l in range(L)),
                       key=lambda cl: ((nc - 1 - cl[0]) + (L - 1 - cl[1]), -cl[1])):
        t0, t1 = chunks[c]
        n = t1 - t0
        ss = s_sweep[l] if piped else main
        sg = s_gemm[l] if piped else main
        with torch.cuda.stream(ss):
            if piped:
                if l < L - 1:
                    ss.wait_event(ev_dx[l + 1][c])
                if c == nc - 1:
                    ss.wait_event(ev_wt[l])
            else:
                if c == nc - 1:
                    N.check(lib.b2t_transpose_f32(_p(prm.w_hh[l]), _p(whh_ts[l]), 3 * H, H, _stream()),
                            "b2t_transpose_f32")
            if ctx.rnn_drop > 0 and l < L - 1:   # gradient through the inter-layer dropout mask
                dropout(dYs[l], dYs[l], n * B * H, ctx.rnn_drop, ctx.seed * 1000003 + 101 + l,
                        elem0=t0 * B * H, x_off=t0 * B * H, y_off=t0 * B * H)
            if c == nc - 1:
                dh_last = _p(dhidden[l].contiguous()) if dhidden is not None else None
            else:
                dh_last = _p(carries[l][(c + 1) % 2])
            dh_out = _p(dh_init[l]) if c == 0 else _p(carries[l][c % 2])
            outb = ctx.outs[l]
            with _Prof("gru_sweep_bwd", 2.0 * n * B * 3 * H * H, n + 1 if mode == 0 else 1):
                N.check(lib.b2t_gru_layer_bwd_f32(
                    C.c_void_p(dYs[l].data_ptr() + 4 * t0 * B * H), dh_last,
                    C.c_void_p(ctx.reserves[l].data_ptr() + 4 * t0 * B * 4 * H),
                    C.c_void_p(outb.data_ptr() + 4 * (1 + t0) * B * H), C.c_void_p(outb.data_ptr() + 4 * t0 * B * H),
                    _p(whh_ts[l]), C.c_void_p(dGs[l].data_ptr() + 4 * t0 * B * 4 * H), dh_out, _p(scratch[l]),
                    n, B, H, sweep_mode_arg(bwd_mode_for(mode), H, "b"), _p(ws.sync_ws(l, Tp, dev, B, H, "b")) if mode >= 1 else None, _stream()),
                    "b2t_gru_layer_bwd_f32")
            if piped:
                ev_bs[l][c] = _ev(ss)
        with torch.cuda.stream(sg):
            if piped:
                sg.wait_event(ev_bs[l][c])
            dx_gemm(l, t0, n)
            if piped:
                ev_dx[l][c] = _ev(sg)
        if piped and c == 0:
            # weight gradients of the whole layer once its last chunk is swept, on their own stream (they overlap
            # the sweeps of the layers below).  Per-chunk accumulation is supported by the helper but measured
            # slower inside the full step, for every layer and also for layer 0 alone (28.2 vs 27.1 ms): more
            # launches competing with the sweeps' CUs.  Layer 0's go to the top layer's GEMM stream (idle by then)
            # so that they overlap the day-layer backward instead of queueing in front of it.
            swg = s_gemm[L - 1] if (l == 0 and L > 1) else s_wg[l]
            if PIPELINE["defer_wgrad"] and l > 0:
                deferred.append((swg, l))     # experiment: weight gradients only once every sweep has finished
            else:
                with torch.cuda.stream(swg):
                    swg.wait_event(ev_bs[l][c])
                    _layer_weight_grads(dims, grd, ctx, ws, dGs, l, M, bucket_cb)
        if (not piped) and c == 0:
            _layer_weight_grads(dims, grd, ctx, ws, dGs, l, M, bucket_cb)

    for swg, l in deferred:
        with torch.cuda.stream(swg):
            swg.wait_event(ev_bs[0][0])
            _layer_weight_grads(dims, grd, ctx, ws, dGs, l, M, bucket_cb)
    # layer-0 input gradient -> day layer (on layer 0's GEMM stream: its dU/dV GEMMs are already ordered there)
    with torch.cuda.stream(s_gemm[0] if piped else main):
        if dims.patch > 0:
            N.check(lib.b2t_patch_fold_f32(_p(dV), _p(dU), B, T, F, Tp, dims.patch, dims.stride, _stream()),
                    "b2t_patch_fold_f32")
        if ctx.in_drop > 0:
            dropout(dU, dU, dU.numel(), ctx.in_drop, ctx.seed * 1000003 + 17)
        # softsign backward in place: dpre = dU * (1-|U|)^2
        N.check(lib.b2t_softsign_bwd_f32(_p(ctx.U), _p(dU), dU.numel(), _stream()), "b2t_softsign_bwd_f32")
        # per-sample partial day gradients, then deterministic reduction by day
        slab = ws.get("day_slab", (B, F, F), dev)
        gemm(ctx.x, dU, slab, M=F, N_=F, K=T, Z=B, a_kc=0, a_s0=F, a_sz=T * F, b_kc=0, b_s0=F, b_sz=T * F, c_s0=F,
             c_sz=F * F)
        N.check(lib.b2t_day_reduce_f32(_p(slab), _p(ctx.day_idx), B, F * F, _p(grd.day_w), grd.day_w_stride, _stream()),
                "b2t_day_reduce_f32")
        bslab = ws.get("day_bslab", (B, pad_to(F, 4)), dev)
        colsum(dU, T, F, F, bslab, Z=B, x_sz=T * F, out_sz=bslab.shape[1])
        N.check(lib.b2t_day_reduce_f32(_p(bslab), _p(ctx.day_idx), B, bslab.shape[1], _p(grd.day_b), grd.day_b_stride,
                                       _stream()), "b2t_day_reduce_f32")
        if bucket_cb:
            bucket_cb("day")
    def h0_grad():
        # h0 gradient: sum over layers and batch rows of the carry after t=0 (rnn_model.py:86,123)
        if not ctx.custom_states:
            colsum(dh_init, L * B, H, H, grd.h0)
        else:
            grd.h0.zero_()
        if bucket_cb:
            bucket_cb("h0")

    if piped and L > 2:
        # Every sweep stream's last launch is followed by a GEMM on that layer's GEMM stream (which waits for it), and
        # the weight-gradient streams are GEMM streams: joining the L GEMM streams joins everything.  The command
        # processor takes ~50 us per barrier packet even when its event has long fired, so the streams that finish
        # early (layers 1 .. L-2) are joined into one of them while the last two are still busy; the main stream then
        # waits for three events instead of L.  The h0 reduction rides on that idle stream (layer 0's sweep is the last
        # one to finish: after it every layer's dh_init is final).
        with torch.cuda.stream(s_gemm[1]):
            for l in range(2, L - 1):
                s_gemm[1].wait_event(_ev(s_gemm[l]))
            s_gemm[1].wait_event(ev_bs[0][0])
            h0_grad()
        for s in (s_gemm[1], s_gemm[L - 1], s_gemm[0]):
            main.wait_event(_ev(s))
    else:
        if piped:
            for s in s_gemm:
                main.wait_event(_ev(s))
        h0_grad()
    return dh_init if want_dstates else None


def _layer_weight_grads(dims, grd, ctx, ws, dGs, l, M, bucket_cb, t0=0, t1=None, accumulate=0, final=True):
    """dW_hh = dGh^T h_prev, dW_ih = dGi^T in, bias gradients = column sums of dG (layer l), over the time
    rows [t0, t1) (default: all).  In the pipelined plan this is called once per chunk as soon as that chunk's
    sweep has finished (first chunk overwrites, later chunks accumulate — fixed order, deterministic), so the
    weight-gradient GEMMs fill the CUs the sweeps leave idle instead of forming a tail after the last sweep."""
    B, T = ctx.B, ctx.T
    F, H = dims.F, dims.H
    dG = dGs[l]
    dev = dG.device
    t1 = ctx.Tp if t1 is None else t1
    K = (t1 - t0) * B
    a0 = t0 * B * 4 * H
    gemm(dG, ctx.outs[l], grd.w_hh[l], M=3 * H, N_=H, K=K, a_kc=0, a_s0=4 * H, a_off=a0, b_kc=0, b_s0=H,
         b_off=t0 * B * H, c_s0=H, splitk=splitk_for(3 * H, H, K), ws=ws, slab=f"splitk_slab{l}", accumulate=accumulate)
    if l == 0:
        In = dims.In0
        bs1 = dims.stride * F if dims.patch > 0 else F
        kw = dict(b_kc=0, b_div=B, b_s1=bs1, b_s0=T * F)
        inp, in_off = ctx.Ud, t0 * bs1
    else:
        In = H
        kw = dict(b_kc=0, b_s0=H)
        inp, in_off = ctx.outs_d[l - 1], (1 + t0) * B * H   # skip the initial-state slot
    if (2 * H) % 128 == 0 and (3 * H) % 128 == 0:   # dGi^T as ONE operand with a gap along m
        gemm(dG, inp, grd.w_ih[l], M=3 * H, N_=In, K=K, a_kc=0, a_s0=4 * H, a_off=a0, c_s0=In, b_off=in_off,
             splitk=splitk_for(3 * H, In, K), ws=ws, slab=f"splitk_slab{l}", accumulate=accumulate, a_brk=2 * H, a_gap=H, **kw)
    else:
        gemm(dG, inp, grd.w_ih[l], M=2 * H, N_=In, K=K, a_kc=0, a_s0=4 * H, a_off=a0, c_s0=In, b_off=in_off,
             splitk=splitk_for(2 * H, In, K), ws=ws, slab=f"splitk_slab{l}", accumulate=accumulate, **kw)
        gemm(dG, inp, grd.w_ih[l], M=H, N_=In, K=K, a_kc=0, a_s0=4 * H, a_off=a0 + 3 * H, c_s0=In, c_off=2 * H * In,
             b_off=in_off, splitk=splitk_for(H, In, K), ws=ws, slab=f"splitk_slab{l}", accumulate=accumulate, **kw)
    s4 = ws.get(f"s4_{l}", (4 * H,), dev)
    colsum(dG, K, 4 * H, 4 * H, s4, accumulate=accumulate, x_off=a0)     # (s_r, s_z, s_nr, s_n)
    if final:
        grd.b_ih[l][:2 * H].copy_(s4[:2 * H]); grd.b_ih[l][2 * H:].copy_(s4[3 * H:])
        grd.b_hh[l].copy_(s4[:3 * H])
        if bucket_cb:
            bucket_cb(f"layer{l}")


# ------------------------------------------------------------------------------------------------
# CTC / decode
# ------------------------------------------------------------------------------------------------
def ctc_loss(logits: torch.Tensor, targets: torch.Tensor, in_len: torch.Tensor, tgt_len: torch.Tensor,
             want_grad: bool, grad_scale: float, ws: Workspace):
    """loss [B] and (optionally) dlogits [B,T,ldd] — fused log-softmax + CTC (rnn_trainer.py:538-545)."""
    _need(logits, name="logits")
    B, T, Cc = logits.shape
    dev = logits.device
    targets = _need(targets.to(device=dev, dtype=torch.int32).contiguous(), torch.int32, "targets")
    in_len = in_len.to(device=dev, dtype=torch.int32).contiguous()
    tgt_len = tgt_len.to(device=dev, dtype=torch.int32).contiguous()
    S_max = max(1, targets.shape[1])
    if targets.shape[1] == 0:
        targets = torch.zeros((B, 1), dtype=torch.int32, device=dev)
    ldd = pad_to(Cc, 4)
    loss = torch.empty((B,), dtype=torch.float32, device=dev)
    alpha = ws.get("ctc_alpha", (2, B, T, 2 * S_max + 1), dev)   # alpha rows, beta rows
    dl = ws.get("ctc_dlogits", (B, T, ldd), dev) if want_grad else None
    with _Prof("ctc_kernel"):
        N.check(N.load().b2t_ctc_loss_f32(_p(logits), _p(targets), _p(in_len), _p(tgt_len), _p(loss), _p(alpha), _p(dl),
                                          B, T, Cc, S_max, ldd, float(grad_scale), _stream()), "b2t_ctc_loss_f32")
    return loss, dl, ldd


def greedy_decode(logits: torch.Tensor, lens: torch.Tensor):
    """argmax -> collapse repeats -> drop blank (rnn_trainer.py:725-728). Returns ids [B,T], lengths [B], argmax [B,T]."""
    _need(logits, name="logits")
    B, T, Cc = logits.shape
    dev = logits.device
    lens = lens.to(device=dev, dtype=torch.int32).contiguous()
    ids = torch.zeros((B, T), dtype=torch.int32, device=dev)
    ln = torch.empty((B,), dtype=torch.int32, device=dev)
    am = torch.empty((B, T), dtype=torch.int32, device=dev)
    N.check(N.load().b2t_greedy_decode_f32(_p(logits), _p(lens), _p(ids), _p(ln), _p(am), B, T, Cc, _stream()),
            "b2t_greedy_decode_f32")
    return ids, ln, am


def edit_distance(a: torch.Tensor, a_len: torch.Tensor, b: torch.Tensor, b_len: torch.Tensor) -> torch.Tensor:
    """Levenshtein distance per row (torchaudio.functional.edit_distance at rnn_trainer.py:734)."""
    dev = a.device
    a = a.to(torch.int32).contiguous(); b = b.to(device=dev, dtype=torch.int32).contiguous()
    a_len = a_len.to(device=dev, dtype=torch.int32).contiguous(); b_len = b_len.to(device=dev, dtype=torch.int32).contiguous()
    Bn = a.shape[0]
    out = torch.empty((Bn,), dtype=torch.int32, device=dev)
    N.check(N.load().b2t_edit_distance_i32(_p(a), _p(a_len), a.shape[1], _p(b), _p(b_len), b.shape[1], _p(out), Bn,
                                           _stream()), "b2t_edit_distance_i32")
    return out
