"""Host-side operators over libb2t_hip.so: thin typed wrappers (torch tensors in, raw pointers out)
and the forward/backward orchestration of the day-layer -> GRU stack -> head -> CTC path.

PyTorch is used for device memory, streams/events and autograd plumbing only; every arithmetic step is a
call into the C ABI (include/b2t.h).  There is no CPU fallback: tensors must live on the HIP device.

model_forward / model_backward are ONE C-ABI call each (b2t_model_forward / b2t_model_backward): the pipelined
execution plan -- per-layer sweep and GEMM streams, time chunks, event edges -- is issued from C++
(csrc/exec.cpp); the ~250 launches of a training step no longer cross the language boundary one by one.
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


def precision_from_args(args) -> bool:
    """Whether a run configured by the reference's yaml dict computes in the bf16 regime.  The reference wraps its forward
    in torch.autocast(dtype=bfloat16, enabled=args['use_amp']) (rnn_trainer.py:527,704; evaluate_model_helpers.py:90) and
    ships `use_amp: true` (rnn_args.yaml:19), so `use_amp` maps to the bf16 mode here too: bf16 operands on the matrix cores
    for every GEMM and recurrent product, fp32 accumulation / gates / CTC / optimizer / master weights.  Order of precedence:
    args['amd_bf16_matmul'] (explicit True / False: the opt-out is `amd_bf16_matmul: false`), the B2T_AMP environment
    variable when set, then `use_amp`."""
    # any mapping with .get(): the reference's entry scripts hand over an OmegaConf DictConfig, which is not a dict subclass
    get = getattr(args, "get", None)
    if get is not None and get("amd_bf16_matmul") is not None:
        return bool(get("amd_bf16_matmul"))
    env = os.environ.get("B2T_AMP")
    if env is not None:
        return env not in ("0", "", "false", "False")
    return bool(get("use_amp", False)) if get is not None else False


def gemm(A, B, Cm, *, M, N_, K, Z=1, a_kc=1, b_kc=1, a_s0=0, a_s1=0, a_div=0, a_sz=0, b_s0=0, b_s1=0, b_div=0,
         b_sz=0, c_s0=0, c_s1=0, c_div=0, c_sz=0, bias=None, bias_sz=0, b_zmap=None, epilogue=0, accumulate=0,
         a_off=0, b_off=0, c_off=0, splitk=1, ws=None, slab="splitk_slab", _splitk=1, _c_ks=0, a_brk=0, a_gap=0,
         a_sum=None, a_sum_ks=0):
    """C = A.B^T through b2t_gemm_f32.  splitk>1 (needs `ws`): partial products go to a workspace slab and
    are summed deterministically into C by b2t_colsum_f32 (used for the weight gradients, K = T*B)."""
    if splitk > 1:
        if Z != 1 or epilogue != 0 or c_div != 0 or c_s0 != N_:
            raise RuntimeError("split-K gemm supports Z=1, dense row-major C, no epilogue")
        sl = ws.get(slab, (splitk, M * N_), Cm.device)
        gemm(A, B, sl, M=M, N_=N_, K=K, a_kc=a_kc, b_kc=b_kc, a_s0=a_s0, a_s1=a_s1, a_div=a_div, b_s0=b_s0,
             b_s1=b_s1, b_div=b_div, c_s0=N_, bias=bias, a_off=a_off, b_off=b_off, _splitk=splitk, _c_ks=M * N_,
             a_brk=a_brk, a_gap=a_gap, a_sum=a_sum, a_sum_ks=a_sum_ks)
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
    d.a_sum, d.a_sum_ks = (a_sum.data_ptr() if a_sum is not None else None), a_sum_ks   # fp32 tile kernel, a_kc=0: per-slice sums over k of A
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


def cumsum_add(w: torch.Tensor, y: torch.Tensor, axis: int):
    """y += cumsum(w, dim=axis) in place (random-walk augmentation, rnn_trainer.py:464-465)."""
    _need(w, name="w"); _need(y, name="y")
    if w.shape != y.shape:
        raise RuntimeError("cumsum_add: shapes differ")
    axis = axis % w.dim()
    outer = int(np.prod(w.shape[:axis], dtype=np.int64)); inner = int(np.prod(w.shape[axis + 1:], dtype=np.int64))
    N.check(N.load().b2t_cumsum_add_f32(_p(w), _p(y), outer, int(w.shape[axis]), inner, _stream()), "b2t_cumsum_add_f32")


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
        self._desc = None

    def desc(self, dims: "ModelDims"):
        """b2t_model_t over these views (built once: the views point into an arena that never moves)."""
        if self._desc is None:
            self._desc = model_desc(dims, self)
        return self._desc


class Grads(Params):
    """Destination views for parameter gradients (normally views of the model's gradient arena)."""


# GRU sweep mode: 0 = one launch per time step; 1 = persistent single-launch sweep with counter hand-off
# (csrc/gru_persistent.hip).  -1 = choose per call: mode 1 whenever its (H/16) x ceil(B/16) workgroups can be
# co-resident, else 0.  (The granule / 4-row-group / fused-stack variants of round 1 lost to mode 1 inside the full step
# and live in attic/ with their measurements in DESIGN.md.)
GRU_MODE = {"value": int(os.environ.get("B2T_GRU_MODE", "-1"))}
MAX_RESIDENT_WGS = 256   # MI355X: 256 CUs; a persistent sweep needs all its workgroups resident at once
# Number of time chunks the layers are software-pipelined over (1 = layer-by-layer on the caller's stream).
PIPELINE = {"chunks": int(os.environ.get("B2T_CHUNKS", "6")),
            # the backward pass is bound by its GEMMs (0.75 TFLOP next to the sweeps) and prefers fewer, larger launches:
            # 6 / 4 measured 23.2 ms against 23.6 for 6 / 6 (6 / 3: 23.6, 6 / 2: 24.2, 8 / 4: 23.1-23.2); 0 = same as forward
            "chunks_bwd": int(os.environ.get("B2T_CHUNKS_BWD", "4")),
            # bit l: layer l's weight gradients accumulate per chunk (measured slower again under the C++ plan: layer 0 only
            # 24.5 ms, layers 0-1 24.8 ms against 23.6 -- the extra GEMM launches take CUs from the sweeps they run next to)
            "wgrad_chunk_mask": int(os.environ.get("B2T_WGRAD_CHUNK_MASK", "0"))}


def gru_mode_for(B: int, H: int) -> int:
    m = GRU_MODE["value"]
    if m in (0, 1):
        return m
    return 1 if (H // 16) * ((B + 15) // 16) <= MAX_RESIDENT_WGS and H <= 1024 else 0


GRU_BF16 = 0x100   # B2T_GRU_BF16 (include/b2t.h): bf16 operands of the recurrent product, persistent mode 1
GRU_WIDE = 0x200   # B2T_GRU_WIDE: 32 hidden units per workgroup
GRU_LOCAL = 0x400  # B2T_GRU_LOCAL: XCD-local hand-off of the fp32 sweeps
GRU_PARITY = 0x800  # B2T_GRU_PARITY: with GRU_LOCAL, the layer's parity (which XCDs its row groups use)
GRU_PAIRED = 0x1000  # B2T_GRU_PAIRED: exact-fp32 backward sweep with its W_hh^T slice in LDS (512-thread workgroups owning two row groups)
GRU_SET_SHIFT = 13   # B2T_GRU_SET_SHIFT: with GRU_PAIRED, bits 13-14 = the sweep's XCD set
GRU_WAVE = 0x8000    # B2T_GRU_WAVE: the pass's L sweeps as ONE launch, the step-granular layer wavefront (csrc/gru_wave.hip)
# bf16 mode (use_amp): the layer wavefront wherever the library holds the shape (H % 16 == 0, H <= 768, B <= 64, L x H / 16 workgroups
# resident); B2T_WAVE=0 selects the round-5 chunk pipeline (the tests compare the two in one process: read per pass)
WAVE = {"on": os.environ.get("B2T_WAVE", "1") not in ("0", "", "false", "False"),
        # time chunks of the wavefront passes (forward, backward): a launch per chunk, one behind the other; what overlaps is the work
        # NEXT to the sweeps on the CUs they leave free -- layer 0's projection of the next chunk, the weight gradients of the chunk
        # before (B2T_WAVE_CHUNKS="f,b", read per pass)
        "chunks": (1, 1), "dirs": "auto"}
# the exact-fp32 backward sweeps as paired sweeps (B2T_BWD_PAIRED=1; H % 32 == 0, H <= 512, B <= 64 -- other shapes ignore the flag)
PAIRED_BWD = {"on": os.environ.get("B2T_BWD_PAIRED", "0") not in ("0", "", "false", "False")}
# which sweeps (exact fp32 or bf16 operands, H <= 512) hand off through one XCD's L2 ("" none, "f", "b", "fb"; B2T_GRU_LOCAL).  Measured at C2: memory-side
# traffic of a backward sweep launch 660 -> 227 MB (1.38x its algorithmic bytes), forward 179 -> 112 MB, a backward launch
# 935 -> 810-830 us, the step -0.1 ms on two boxes (with WRITE-THROUGH payload stores; ordinary stores cost the GEMMs 0.3-1 ms).
# Ignored where the library's dispatch probe fails.
LOCAL_F32 = {"dirs": os.environ.get("B2T_GRU_LOCAL", "fb")}
# streaming calls (inference, <= 8 output frames, B <= 64) as one fused launch, csrc/stream.hip: opt-in (B2T_STREAM_FUSED=1) --
# measured 151-159 us per frame against 163 through the executor, and a grid-barrier kernel wants the chip to itself
STREAM = {"fused": os.environ.get("B2T_STREAM_FUSED", "0") not in ("0", "", "false", "False"),
          # streaming calls of a model in eval() mode replayed as hipGraphs from their third call on (rnn_model._graph_forward)
          "graph": os.environ.get("B2T_STREAM_GRAPH", "1") not in ("0", "", "false", "False")}
# which sweeps run with 32-unit workgroups under AMP: "" none, "f" forward, "b" backward, "fb" both (B2T_AMP_WIDE;
# measured at C2: 18.4 / 17.5 / 17.1 / 16.1 ms per step)
AMP["wide"] = os.environ.get("B2T_AMP_WIDE", "fb")


# exact-fp32 sweeps with 32 hidden units per workgroup (the fp32 weight slice of 32 units takes 192 of a lane's 512
# registers, one workgroup per CU): "" none, "f" forward, "b" backward, "fb" both (B2T_WIDE_F32)
WIDE_F32 = {"dirs": os.environ.get("B2T_WIDE_F32", "f")}   # measured at C2: forward 23.9 -> 23.3 ms per step, backward no gain


def sweep_mode_arg(mode: int, H: int = 0, direction: str = "f") -> int:
    """`mode` argument of b2t_gru_layer_fwd/bwd_f32: under set_amp(True) the persistent sweeps take bf16 operands."""
    if not (AMP["on"] and AMP.get("sweeps", True) and mode == 1):
        local = GRU_LOCAL if (mode == 1 and direction in LOCAL_F32["dirs"] and H <= 512) else 0
        if mode == 1 and direction in WIDE_F32["dirs"] and H % 32 == 0 and H <= 512:
            return mode | GRU_WIDE | local
        if mode == 1 and direction == "b" and PAIRED_BWD["on"] and local:
            return mode | local | GRU_PAIRED
        return mode | local
    wide = GRU_WIDE if (direction in AMP["wide"] and H % 32 == 0 and H <= int(os.environ.get("B2T_AMP_WIDE_MAXH", "768"))) else 0
    # C2 with bf16 operands: 11.3 -> 11.15 ms per step; with 32-unit workgroups a row group of H = 768 is 24 workgroups and still
    # fits one XCD (B2T_GRU_LOCAL_MAXH, also read by the library)
    local_maxh = int(os.environ.get("B2T_GRU_LOCAL_MAXH", "512")) if wide else 512
    local = GRU_LOCAL if (direction in LOCAL_F32["dirs"] and H <= local_maxh) else 0
    return mode | GRU_BF16 | wide | local


def host_api_probe(n: int = 200) -> dict:
    """Host latency (microseconds, p50 / p90) of the three runtime calls a pass is made of -- a small kernel launch through the C ABI,
    hipEventRecord, hipStreamWaitEvent -- taken at start-up (~3 ms).  On this pool a process sometimes comes up in a mode in which
    every HIP call of its whole life is 2.5-3x slower (NOTES.md 5 / R4.10d / R5.5: host enqueue 4.3-4.7 ms per C2 step instead of
    1.1-1.6, the step 1-2 ms slower); a healthy process measures ~7.3 / 4.7 / 0.5 us here (14 processes under six environments,
    `tools/r5_slowmode.py`).  `slow` = the kernel launch's p50 above 15 us.  The trainer logs the numbers (and a warning when slow);
    bench.py reports them in `box`, so that a run caught in the mode carries its signature."""
    import time
    dev = torch.device("cuda", torch.cuda.current_device())
    a = torch.zeros(64, 64, device=dev); b = torch.zeros(64, 64, device=dev)
    lib = N.load()
    s2, ev = torch.cuda.Stream(), torch.cuda.Event()
    out = {}
    for name, fn in (("kernel_launch_us", lambda: lib.b2t_transpose_f32(_p(a), _p(b), 64, 64, _stream())),
                     ("event_record_us", lambda: ev.record()), ("stream_wait_event_us", lambda: s2.wait_event(ev))):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        ts = []
        for i in range(n):
            t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
            if i % 50 == 49:
                torch.cuda.synchronize()
        ts.sort()
        out[name] = dict(p50=round(ts[len(ts) // 2] * 1e6, 2), p90=round(ts[int(len(ts) * 0.9)] * 1e6, 2))
    # ... and the same calls the way a pass issues them: four streams, every launch behind an event of the stream before it, nothing
    # waited for -- microseconds per runtime call with the queues filling up.  (Round 5 caught one slow process: its IDLE latencies
    # above were those of a healthy one, 7.5 / 4.7 / 0.5 us, while its step enqueued in 5.4 ms instead of 1.5 -- the mode shows
    # under load only.)
    streams = [torch.cuda.Stream() for _ in range(4)]
    evs = [torch.cuda.Event() for _ in range(4)]
    cur = torch.cuda.current_stream()
    ncall = 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    import ctypes as _C
    handles = [_C.c_void_p(st.cuda_stream) for st in streams]
    for rnd in range(48):
        for i, st in enumerate(streams):
            if rnd or i:
                st.wait_event(evs[(i - 1) % 4])
            lib.b2t_transpose_f32(_p(a), _p(b), 64, 64, handles[i])
            lib.b2t_transpose_f32(_p(b), _p(a), 64, 64, handles[i])
            evs[i].record(st)
            ncall += 4
    t_burst = time.perf_counter() - t0
    cur.wait_event(evs[3])
    torch.cuda.synchronize()
    out["burst_us_per_call"] = round(t_burst / ncall * 1e6, 2)      # recorded, not judged: no slow process has left this number yet
    out["slow"] = bool(out["kernel_launch_us"]["p50"] > 15.0)
    return out


def slow_mode_probe(hops: int = 64) -> dict:
    """What a process that found itself in the pool's slow-host mode measures about itself (round-5 verdict item 1b): the
    cross-queue hop the plan is made of -- a chain of tiny launches that alternates between two streams through events: microseconds
    per hop on the GPU timeline and of host time --, the burst latency of host_api_probe, and the process's scheduling state (CPU it
    runs on, that CPU's clock, context switches during the chain, load average, NUMA node of the CPU against the GPU's)."""
    import time
    dev = torch.device("cuda", torch.cuda.current_device())
    lib = N.load()
    a = torch.zeros(64, 64, device=dev); b = torch.zeros(64, 64, device=dev)
    s0, s1 = torch.cuda.current_stream(), torch.cuda.Stream()
    h = [C.c_void_p(s0.cuda_stream), C.c_void_p(s1.cuda_stream)]
    evs = [torch.cuda.Event() for _ in range(hops)]

    def ctx():
        try:
            d = dict(l.split(":", 1) for l in open("/proc/self/status").read().splitlines() if "ctxt_switches" in l)
            return int(d["voluntary_ctxt_switches"]), int(d["nonvoluntary_ctxt_switches"])
        except Exception:
            return None
    out = {}
    for rep in range(2):
        torch.cuda.synchronize()
        c0 = ctx()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(s0)
        for i in range(hops):
            src, dst = (s0, s1) if i % 2 == 0 else (s1, s0)
            lib.b2t_transpose_f32(_p(a), _p(b), 64, 64, h[i % 2])
            evs[i].record(src)
            dst.wait_event(evs[i])
        if hops % 2:
            s0.wait_event(evs[-1])
        e1.record(s0)
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        c1 = ctx()
        out = dict(hop_us_gpu=round(e0.elapsed_time(e1) * 1e3 / hops, 2), hop_us_host=round(t_host * 1e6 / hops, 2),
                   ctxt_switches_during_chain=(None if not (c0 and c1) else [c1[0] - c0[0], c1[1] - c0[1]]))
    try:
        cpu = int(open("/proc/self/stat").read().rsplit(")", 1)[1].split()[36])
        out["cpu"] = cpu
        for name, path in (("cpu_khz", f"/sys/devices/system/cpu/cpu{cpu}/cpufreq/scaling_cur_freq"),):
            try:
                out[name] = int(open(path).read())
            except Exception:
                pass
        import glob
        nodes = [os.path.basename(g) for g in glob.glob(f"/sys/devices/system/cpu/cpu{cpu}/node*")]
        out["cpu_numa_node"] = nodes[0] if nodes else None
        pr = torch.cuda.get_device_properties(dev)
        bus = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        try:
            out["gpu_numa_node"] = int(open(f"/sys/bus/pci/devices/{bus}/numa_node").read())
        except Exception:
            pass
    except Exception:
        pass
    try:
        out["loadavg"] = [round(v, 2) for v in os.getloadavg()]
        out["threads"] = len(os.listdir("/proc/self/task"))
        out["affinity"] = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        out["host_api_us"] = host_api_probe(100)
    except Exception:
        pass
    return out


def gru_sync_check(sync_ws, T: int, B: int):
    """Raise if the last persistent sweep on sync_ws reported a hand-off timeout (synchronises)."""
    st = C.c_int(0)
    N.check(N.load().b2t_gru_sync_status(_p(sync_ws), T, B, C.byref(st), _stream()), "b2t_gru_sync_status")
    if st.value != 0:
        raise RuntimeError("persistent GRU sweep: inter-workgroup hand-off timed out (results invalid)")


def _crowded(B: int, H: int) -> bool:
    """Two sweeps of this shape cannot be resident together (see time_chunks)."""
    wgs = (H // 16) * ((B + 15) // 16)
    return 2 * wgs > MAX_RESIDENT_WGS * (2 if H <= 512 else 1)


def time_chunks(Tp: int, B: int = 64, H: int = 512, amp: bool = False) -> int:
    """Number of time chunks of the forward layer pipeline.  Pipelining pays only when the sweeps of two layers can be
    resident together: a sweep is (H/16) * ceil(B/16) workgroups that must all run at once, and the chip holds 2 backward-sweep
    workgroups per CU up to H = 512 but only 1 beyond (the W_hh slice takes the whole register file).  At H = 768, B = 64
    (192 workgroups of 256 slots) concurrent fp32 sweeps just block each other: 37-93 ms per step with 6 chunks against
    19.8 ms with the layers in sequence (tools/bench_c3.py).  With bf16 operands (amp) the slices are half as large, the
    sweeps run with 32-unit workgroups (96 per sweep) and the GEMMs are short: 3 forward / 3 backward chunks (2 backward until round 5) let the four
    queues overlap GEMMs and sweeps of different layers (shipped shape: 9.7 ms serial, 7.8 at 2 / 2, 7.4-7.5 at 3 / 2 and
    4 / 2, 7.9 at 6 / 4)."""
    if _crowded(B, H) and "B2T_CHUNKS" not in os.environ:
        wgs = (H // 16) * ((B + 15) // 16)
        return 3 if (amp and AMP["sweeps"] and wgs <= MAX_RESIDENT_WGS and Tp >= 48) else 1
    return max(1, min(PIPELINE["chunks"], Tp // 16))


def time_chunks_bwd(Tp: int, B: int, H: int, amp: bool, fwd_chunks: int) -> int:
    """Time chunks of the backward pass (0: the plan runs in sequence, as the forward does with one chunk)."""
    if fwd_chunks == 1:
        return 0
    if _crowded(B, H) and "B2T_CHUNKS_BWD" not in os.environ:
        # (only the bf16 mode gets here with fwd_chunks > 1.)  2 until the end of round 5; re-swept on that round's kernels: 3 / 3
        # 5.64-5.67 ms against 3 / 2 5.70-5.74 (three pairs in one call), 4 / 3 5.80, 4 / 2 5.91, 2 / 2 6.01
        return max(2, min(3, fwd_chunks, Tp // 16))
    return max(0, min(PIPELINE["chunks_bwd"], fwd_chunks if "B2T_CHUNKS_BWD" not in os.environ else PIPELINE["chunks_bwd"], Tp // 16))


BUCKET_NAMES = lambda L: ["head"] + [f"layer{l}" for l in range(L)] + ["h0", "day"]   # ids of b2t_bucket_cb
PROF_KINDS = ["gemm_f32_kernel<0,0>", "gemm_f32_kernel<0,1>", "gemm_f32_kernel<1,0>", "gemm_f32_kernel<1,1>",
              "gemm_bf16_kernel<0,0>", "gemm_bf16_kernel<0,1>", "gemm_bf16_kernel<1,0>", "gemm_bf16_kernel<1,1>",
              "gru_sweep_fwd", "gru_sweep_bwd"]   # kinds of b2t_exec_profile_read (one name per rocprof symbol)


class Workspace:
    """Per-model execution state: the C++ executor (streams + events of the pipelined plan, csrc/exec.cpp), its
    persistent sync block (sweep counters + sticky error words, zeroed once) and the per-shape pass workspaces
    (allocated once per shape from torch's caching allocator)."""
    def __init__(self):
        self.bufs: Dict[Tuple, torch.Tensor] = {}
        self._exec = None
        self._exec_L = 0

    def get(self, name, shape, device, dtype=torch.float32):
        key = (name, tuple(shape), str(device), dtype)
        t = self.bufs.get(key)
        if t is None:
            t = (torch.zeros if dtype == torch.int32 else torch.empty)(shape, dtype=dtype, device=device)
            self.bufs[key] = t
        return t

    def executor(self, L: int):
        if self._exec is None or self._exec_L < L:
            if self._exec is not None:
                N.load().b2t_exec_destroy(self._exec)
            h = C.c_void_p()
            N.check(N.load().b2t_exec_create(L, C.byref(h)), "b2t_exec_create")
            self._exec, self._exec_L = h, L
        return self._exec

    def __del__(self):
        try:
            if self._exec is not None:
                N.load().b2t_exec_destroy(self._exec)
        except Exception:
            pass

    def sync(self, L: int, device) -> torch.Tensor:
        """[2L][words] int32: block l = forward sweep of layer l, block L + l = its backward sweep; word 0 of a block is
        the sweep's sticky error word."""
        words = N.load().b2t_gru_sync_bytes(0) // 4
        return self.get("exec_sync", (2 * L + 1, words), device, torch.int32)   # last block: tile counters of the split-K GEMMs (b2t_exec_sync_bytes)

    def error_words(self, L: int, device):
        """(pointer, count, stride in words) of the sweeps' error words, for b2t_grad_norm_clip_f32."""
        t = self.sync(L, device)
        return t, 2 * L, t.shape[1]

    def check_sync(self):
        """Raise if any persistent sweep that used this workspace reported a hand-off timeout (synchronises)."""
        for key, buf in self.bufs.items():
            if key[0] == "exec_sync" and int(buf[:, 0].abs().max().item()) != 0:
                raise RuntimeError("persistent GRU sweep: inter-workgroup hand-off timed out — results invalid")

    def graph_stats(self):
        """(graphs built, passes replayed, a build failed) of this workspace's executor under B2T_EXEC_GRAPH=1."""
        if self._exec is None:
            return 0, 0, False
        import ctypes as C
        b, r, f = C.c_longlong(0), C.c_longlong(0), C.c_int(0)
        N.check(N.load().b2t_exec_graph_stats(self._exec, C.byref(b), C.byref(r), C.byref(f)), "b2t_exec_graph_stats")
        return int(b.value), int(r.value), bool(f.value)

    def profile(self, on: bool):
        if self._exec is not None:
            N.check(N.load().b2t_exec_profile(self._exec, 1 if on else 0), "b2t_exec_profile")

    def profile_read(self):
        """[(name, flops, launches, seconds)] of the executor's launches since profile(True) (synchronises)."""
        if self._exec is None:
            return []
        cap = 1 << 16
        kind = (C.c_int * cap)(); fl = (C.c_double * cap)(); ms = (C.c_float * cap)()
        n = N.load().b2t_exec_profile_read(self._exec, kind, fl, ms, cap)
        if n < 0:
            raise RuntimeError("b2t_exec_profile_read failed: " + N.last_error())
        return [(PROF_KINDS[kind[i]], fl[i], 1, ms[i] * 1e-3) for i in range(n)]


def model_desc(dims: "ModelDims", prm: "Params") -> "N.ModelDesc":
    d = N.ModelDesc()
    d.F, d.H, d.D, d.C, d.L, d.patch, d.stride = dims.F, dims.H, dims.D, dims.C, dims.L, dims.patch, dims.stride
    d.day_w, d.day_b = prm.day_w.data_ptr(), prm.day_b.data_ptr()
    d.day_w_stride, d.day_b_stride = prm.day_w_stride, prm.day_b_stride
    for l in range(dims.L):
        d.w_ih[l], d.w_hh[l] = prm.w_ih[l].data_ptr(), prm.w_hh[l].data_ptr()
        d.b_ih[l], d.b_hh[l] = prm.b_ih[l].data_ptr(), prm.b_hh[l].data_ptr()
    d.out_w, d.out_b, d.h0 = prm.out_w.data_ptr(), prm.out_b.data_ptr(), prm.h0.data_ptr()
    return d


class ForwardCtx:
    pass


# ------------------------------------------------------------------------------------------------
# forward / backward: one C-ABI call each (csrc/exec.cpp issues the launches of the plan)
# ------------------------------------------------------------------------------------------------
def copy_segments(pairs):
    """[(src, dst)] contiguous 32-bit tensors of equal size, at most four: all copied by ONE launch on the current stream."""
    n = len(pairs)
    src = (C.c_void_p * n)(*[p[0].data_ptr() for p in pairs]); dst = (C.c_void_p * n)(*[p[1].data_ptr() for p in pairs])
    words = (C.c_longlong * n)(*[p[0].numel() for p in pairs])
    for s_, d_ in pairs:
        if s_.numel() != d_.numel() or s_.element_size() != 4 or d_.element_size() != 4 or not (s_.is_contiguous() and d_.is_contiguous()):
            raise RuntimeError("copy_segments: contiguous 32-bit tensors of equal size expected")
    N.check(N.load().b2t_copy_segments_b32(src, dst, words, n, _stream()), "b2t_copy_segments_b32")


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
    if day_idx.numel() != B:
        raise RuntimeError("day_idx must have one entry per batch row")
    dev = x.device
    md = prm.desc(dims)
    # A streaming call -- inference on a handful of patch frames with carried state (the online decoder's frame-by-frame use,
    # BASELINE configs[4]) -- as ONE launch (csrc/stream.hip) instead of the executor's ~12 dependent ones: opt-in, fp32 only.
    if (not save) and STREAM["fused"] and not AMP["on"] and in_drop == 0.0 and rnn_drop == 0.0 and \
            lib.b2t_stream_supported(C.byref(md), B, T):
        nb = lib.b2t_stream_ws_bytes(C.byref(md), B, T)
        buf = ws.get("stream_ws", ((nb + 3) // 4,), dev)
        sync = ws.get("stream_sync", (lib.b2t_stream_sync_bytes() // 4,), dev, torch.int32)
        logits = torch.empty((B, Tp, Cc), dtype=torch.float32, device=dev)
        hidden = torch.empty((L, B, H), dtype=torch.float32, device=dev)
        if states is not None:
            _need(states, name="states")
            if tuple(states.shape) != (L, B, H):
                raise RuntimeError(f"states must be [{L},{B},{H}]")
        N.check(lib.b2t_stream_forward_f32(C.byref(md), B, T, _p(x), _p(day_idx), _p(states), _p(logits), _p(hidden), _p(buf),
                                           nb, _p(sync), _stream()), "b2t_stream_forward_f32")
        return logits, hidden, None
    mode = gru_mode_for(B, H)
    ps = N.PassDesc()
    ps.B, ps.T, ps.chunks = B, T, time_chunks(Tp, B, H, AMP["on"])
    ps.fwd_mode, ps.bwd_mode = sweep_mode_arg(mode, H, "f"), sweep_mode_arg(mode, H, "b")
    ps.bf16_gemm, ps.save = int(AMP["on"]), int(bool(save))
    ps.in_drop, ps.rnn_drop, ps.seed = float(in_drop), float(rnn_drop if L > 1 else 0.0), int(seed) & (2 ** 64 - 1)
    ps.chunks_bwd = time_chunks_bwd(Tp, B, H, AMP["on"], ps.chunks)
    ps.wgrad_chunk_mask = PIPELINE["wgrad_chunk_mask"]
    # (not for streaming-sized calls: a launch loads 192-288 registers of weights per wave before its first step)
    wave_form = lib.b2t_gru_wave_supported(L, Tp, B, H) if AMP["on"] and mode == 1 else 0     # 0 no, 1 the 16-unit form, 2 the K-split form
    if AMP["on"] and AMP.get("sweeps", True) and mode == 1 and WAVE["on"] and os.environ.get("B2T_WAVE", "1") != "0" and Tp >= 16 and wave_form:
        wc = os.environ.get("B2T_WAVE_CHUNKS")
        # chunks (forward launches, backward consumer chunks): 1, 1 -- forward launches per chunk measured equal (C2: 8.98 / 9.01 /
        # 8.98 ms with 1 / 2 / 3), gated backward consumers worse with every chunk (NOTES.md R6.2)
        cf, cb = (int(v) for v in wc.split(",")) if wc else WAVE["chunks"]
        # Which passes.  Measured (NOTES.md R6.2): the FORWARD wavefront beats the chunk pipeline at both bench shapes (C2 9.80 ->
        # 8.98 ms, shipped shape 5.61 -> 5.11); the backward one (8 us per step at C2, 10.4 at the shipped shape: three times the
        # forward's operand bytes per CU, and its weight gradients then run behind it instead of beside it) does not (9.98 / 5.29) --
        # so the default is "f" (B2T_WAVE_DIRS=fb / b: measurement knob; the backward kernel stays tested)
        # -- in the 16-unit form.  In the K-split form (H % 128 == 0, H <= 512: half the operand bytes per CU, data-polled hand-off)
        # the backward wavefront wins as well (C2: 8.33 forward only, 8.05 both): "auto" = both passes there
        dirs = os.environ.get("B2T_WAVE_DIRS", WAVE.get("dirs", "auto"))
        if dirs == "auto":
            dirs = "fb" if wave_form == 2 else "f"
        if "f" in dirs:
            ps.fwd_mode |= GRU_WAVE
            ps.chunks = max(1, min(cf, Tp // 16))
        if "b" in dirs:
            ps.bwd_mode |= GRU_WAVE
            ps.chunks_bwd = max(1, min(cb, Tp // 16))
    nbytes = lib.b2t_pass_ws_bytes(C.byref(md), C.byref(ps))
    if nbytes == 0:
        raise RuntimeError("b2t_pass_ws_bytes: bad model / pass description")
    nfl = (nbytes + 3) // 4
    # the pass workspace holds what backward needs: workspace-owned (one per shape) when the caller guarantees one
    # forward/backward in flight (the trainer's fused step) or nothing is saved; a fresh allocation otherwise
    if (not save) or reuse_saved:
        buf = ws.get("pass_ws", (nfl,), dev)
    else:
        buf = torch.empty((nfl,), dtype=torch.float32, device=dev)
    logits = torch.empty((B, Tp, Cc), dtype=torch.float32, device=dev)
    hidden = torch.empty((L, B, H), dtype=torch.float32, device=dev)
    if states is not None:
        _need(states, name="states")
        if tuple(states.shape) != (L, B, H):
            raise RuntimeError(f"states must be [{L},{B},{H}]")
    sync = ws.sync(L, dev)
    N.check(lib.b2t_model_forward(ws.executor(L), C.byref(md), C.byref(ps), _p(x), _p(day_idx), _p(states), _p(logits),
                                  _p(hidden), _p(buf), _p(sync), _stream()), "b2t_model_forward")
    if not save:
        return logits, hidden, None
    ctx = ForwardCtx()
    ctx.x, ctx.day_idx, ctx.buf, ctx.ps = x, day_idx, buf, ps
    ctx.B, ctx.T, ctx.Tp = B, T, Tp
    ctx.custom_states = states is not None
    return logits, hidden, ctx


def model_backward(dims: ModelDims, prm: Params, grd: Grads, ctx: ForwardCtx, dlogits: torch.Tensor, ldd: int,
                   ws: Workspace, dhidden: Optional[torch.Tensor] = None, want_dstates: bool = False,
                   bucket_cb=None):
    """Gradients of every parameter given dlogits [B,T',ldd] (ldd multiple of 4, >= C).
    Overwrites the destinations in `grd` (days absent from the batch are not touched).
    Follows SURVEY Appendix A2/A3; replaces loss.backward() at rnn_trainer.py:547.
    bucket_cb(name) is called (with the producing stream current) as soon as a gradient bucket is completely enqueued:
    "head", "layer{l}", "h0", "day" — the data-parallel reducer hooks its all-reduce there."""
    lib = N.load()
    L = dims.L
    dev = dlogits.device
    _need(dlogits, name="dlogits")
    names = BUCKET_NAMES(L)
    err: List[BaseException] = []

    def _cb(user, bucket, stream):
        try:
            # The producing stream as a torch stream.  A NULL hipStream_t (ctypes hands it over as None) is torch's DEFAULT
            # stream -- torch.cuda.ExternalStream(0) is NOT: events recorded on it are not ordered after work launched on
            # stream 0 (measured: tools/stream_probe.py; an all-reduce hooked there read the head gradients before the
            # kernels producing them had run -- found by tests/test_gpu_dp_procs.py).
            prod = torch.cuda.ExternalStream(int(stream), device=dev) if stream else torch.cuda.default_stream(dev)
            with torch.cuda.stream(prod):
                bucket_cb(names[bucket])
        except BaseException as e:   # never unwind through the C frames
            err.append(e)

    cb = N.BUCKET_CB(_cb) if bucket_cb is not None else N.BUCKET_CB()
    dstates = torch.empty((L, ctx.B, dims.H), dtype=torch.float32, device=dev) if want_dstates else None
    if dhidden is not None:
        dhidden = _need(dhidden.contiguous(), name="dhidden")
    N.check(lib.b2t_model_backward(ws.executor(L), C.byref(prm.desc(dims)), C.byref(grd.desc(dims)), C.byref(ctx.ps),
                                   _p(ctx.x), _p(ctx.day_idx), _p(dlogits), int(ldd), _p(dhidden), _p(dstates),
                                   int(ctx.custom_states), _p(ctx.buf), _p(ws.sync(L, dev)), cb, None, _stream()),
            "b2t_model_backward")
    if err:
        raise err[0]
    return dstates


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
