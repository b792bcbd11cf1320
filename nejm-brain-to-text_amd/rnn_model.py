"""GRUDecoder — MI355X-native drop-in for the reference's model class.

Same constructor, `forward(x, day_idx, states=None, return_state=False)` signature, parameter names
and state_dict layout as model_training/rnn_model.py:4-134 (so the pretrained t15 checkpoint loads
and `train_model.py` / `evaluate_model.py` style scripts run unchanged), but the arithmetic is the
hand-written HIP path behind include/b2t.h:

    day layer   : per-sample GEMM indexed by day (no [B,512,512] gather), fused bias + softsign
    patching    : implicit im2col — overlapping rows of the batch-first activation (lda = stride*F)
    GRU stack   : fp32-MFMA input projections + recurrent sweep kernels (csrc/gru*.hip)
    head        : GEMM writing batch-first logits

All parameters live in ONE contiguous fp32 arena (each tensor padded to 1024 floats) with a matching
gradient arena, so gradient clipping, AdamW and the data-parallel all-reduce are single launches /
single collectives over flat memory.  There is no CPU fallback: forward() requires the HIP device.
"""
from __future__ import annotations

import math
from typing import List, Optional

import torch
from torch import nn

import b2t_ops as ops


class _GRUParams(nn.Module):
    """Parameter container with torch.nn.GRU's names, shapes, registration order and default
    initialisation (uniform(-1/sqrt(H), 1/sqrt(H)) over every tensor in registration order), so that
    seeding reproduces the reference's initial weights (rnn_model.py:65-79)."""

    def __init__(self, input_size: int, hidden_size: int, num_layers: int, dropout: float):
        super().__init__()
        self.input_size, self.hidden_size, self.num_layers, self.dropout = input_size, hidden_size, num_layers, dropout
        for l in range(num_layers):
            in_l = input_size if l == 0 else hidden_size
            self.register_parameter(f"weight_ih_l{l}", nn.Parameter(torch.empty(3 * hidden_size, in_l)))
            self.register_parameter(f"weight_hh_l{l}", nn.Parameter(torch.empty(3 * hidden_size, hidden_size)))
            self.register_parameter(f"bias_ih_l{l}", nn.Parameter(torch.empty(3 * hidden_size)))
            self.register_parameter(f"bias_hh_l{l}", nn.Parameter(torch.empty(3 * hidden_size)))
        stdv = 1.0 / math.sqrt(hidden_size) if hidden_size > 0 else 0
        for w in self.parameters():
            nn.init.uniform_(w, -stdv, stdv)


class _ModelFn(torch.autograd.Function):
    """Autograd bridge: forward/backward both run the HIP path; parameter gradients are returned to
    autograd as ordinary tensors (so `loss.backward()` behaves like the reference)."""

    @staticmethod
    def forward(ctx, model, x, day_idx, states, want_grad, *params):
        logits, hidden, fctx = ops.model_forward(model._dims, model._kernel_params(), x, day_idx, states, model._ws,
                                                 save=want_grad, in_drop=model._p_in(), rnn_drop=model._p_rnn(),
                                                 seed=model._next_seed())
        ctx.model, ctx.fctx = model, fctx
        ctx.set_materialize_grads(False)
        return logits, hidden

    @staticmethod
    def backward(ctx, dlogits, dhidden):
        model, fctx = ctx.model, ctx.fctx
        if fctx is None:
            raise RuntimeError("GRUDecoder forward ran without saving activations (torch.no_grad?)")
        dims = model._dims
        ldd = ops.pad_to(dims.C, 4)
        dev = fctx.x.device
        dl = torch.zeros((fctx.B, fctx.Tp, ldd), dtype=torch.float32, device=dev)
        if dlogits is not None:
            dl[:, :, :dims.C] = dlogits
        grd, holders = model._fresh_grads(dev, fctx.day_idx)
        dh = dhidden.contiguous() if dhidden is not None else None
        dstates = ops.model_backward(dims, model._kernel_params(), grd, fctx, dl, ldd, model._ws, dhidden=dh,
                                     want_dstates=fctx.custom_states)
        grads = model._collect_grads(holders, fctx.day_idx)
        return (None, None, None, dstates.clone() if dstates is not None else None, None) + tuple(grads)


class GRUDecoder(nn.Module):
    """Day-specific input layers + stacked GRU + linear head (see module docstring)."""

    def __init__(self, neural_dim, n_units, n_days, n_classes, rnn_dropout=0.0, input_dropout=0.0, n_layers=5,
                 patch_size=0, patch_stride=0):
        super().__init__()
        self.neural_dim, self.n_units, self.n_classes = neural_dim, n_units, n_classes
        self.n_layers, self.n_days = n_layers, n_days
        self.rnn_dropout, self.input_dropout = rnn_dropout, input_dropout
        self.patch_size, self.patch_stride = patch_size, patch_stride

        # identity day matrices / zero biases (rnn_model.py:50-55)
        self.day_weights = nn.ParameterList([nn.Parameter(torch.eye(neural_dim)) for _ in range(n_days)])
        self.day_biases = nn.ParameterList([nn.Parameter(torch.zeros(1, neural_dim)) for _ in range(n_days)])

        self.input_size = neural_dim * patch_size if patch_size > 0 else neural_dim
        self.gru = _GRUParams(self.input_size, n_units, n_layers, rnn_dropout)
        # orthogonal recurrent / xavier input weights, visited in registration order (rnn_model.py:75-79)
        for name, param in self.gru.named_parameters():
            if "weight_hh" in name:
                nn.init.orthogonal_(param)
            if "weight_ih" in name:
                nn.init.xavier_uniform_(param)
        self.out = nn.Linear(n_units, n_classes)
        nn.init.xavier_uniform_(self.out.weight)
        self.h0 = nn.Parameter(nn.init.xavier_uniform_(torch.zeros(1, 1, n_units)))

        self._dims = ops.ModelDims(neural_dim, n_units, n_days, n_classes, n_layers, patch_size, patch_stride)
        self._ws = ops.Workspace()
        self._graphs = {}          # streaming shapes replayed as hipGraphs (_graph_forward)
        self._expect = None        # [(parameter dict, key, the tensor's address inside the arena)] (_arena_ok)
        self._owners = []
        self._arena: Optional[torch.Tensor] = None
        self._grad_arena: Optional[torch.Tensor] = None
        self._layout = None
        self._seed_base = int(torch.initial_seed()) & 0x7FFFFFFF
        self._seed_ctr = 0
        self._kp = None

    # ------------------------------------------------------------------ arena ------------------
    def _param_order(self):
        L = self.n_layers
        order = [(f"day_weights.{i}", self.day_weights[i]) for i in range(self.n_days)]
        order += [(f"day_biases.{i}", self.day_biases[i]) for i in range(self.n_days)]
        for l in range(L):
            for nm in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
                order.append((f"gru.{nm}_l{l}", getattr(self.gru, f"{nm}_l{l}")))
        order += [("out.weight", self.out.weight), ("out.bias", self.out.bias), ("h0", self.h0)]
        return order

    def _arena_ok(self) -> bool:
        """Every Parameter still is the view into the arena that pack() made it (a .to(), a `p.data = ...` or a replaced
        Parameter object breaks that).  Runs on every forward: the expected addresses are kept in a flat list so that the check
        is one data_ptr() comparison per tensor (113 tensors: ~15 us; building the name list each time cost 60-80 us, which was
        a third of a streaming call)."""
        if self._arena is None:
            return False
        exp = self._expect
        if exp is None:
            return False
        if self._parameters["h0"].device != self._arena.device:
            return False
        for name, owner in self._owners:          # a replaced submodule (model.gru = ...) brings parameters that are not in the arena
            if getattr(self, name) is not owner:
                return False
        for params, key, want in exp:
            if params[key].data_ptr() != want:
                return False
        return True

    def _build_expect(self):
        base = self._arena.data_ptr()
        exp, owners = [], {}
        for (name, _), (off, n) in zip(self._param_order(), self._layout["spans"]):
            mod_name, _, leaf = name.rpartition(".")
            owner = getattr(self, mod_name) if mod_name else self      # nn.ParameterList / nn.GRU / nn.Linear / the model itself
            if mod_name:
                owners[mod_name] = owner
            exp.append((owner._parameters, leaf, base + 4 * off))
        self._expect, self._owners = exp, list(owners.items())

    def pack(self, device=None):
        """(Re)build the parameter arena on `device` and re-point every Parameter's storage into it.
        Called lazily by forward(); call explicitly after .to()/load_state_dict() if you hold views."""
        order = self._param_order()
        device = device or order[0][1].device
        spans, off = [], 0
        for name, p in order:
            n = p.numel()
            spans.append((off, n))
            off += ops.pad_to(n)
        arena = torch.zeros((off,), dtype=torch.float32, device=device)
        for (name, p), (o, n) in zip(order, spans):
            arena[o:o + n].copy_(p.detach().reshape(-1).to(device=device, dtype=torch.float32))
            p.data = arena[o:o + n].view(p.shape)
        self._arena = arena
        self._grad_arena = torch.zeros_like(arena)
        self._layout = dict(spans=spans, names=[n for n, _ in order], total=off)
        self._kp = None
        self._build_expect()
        return self

    def arena(self):
        if not self._arena_ok():
            self.pack()
        return self._arena

    def grad_arena(self):
        self.arena()
        return self._grad_arena

    def layout(self):
        self.arena()
        return self._layout

    def _span(self, name):
        lay = self.layout()
        return lay["spans"][lay["names"].index(name)]

    def _views(self, arena):
        F, H, L, D = self.neural_dim, self.n_units, self.n_layers, self.n_days
        def v(name, shape=None):
            o, n = self._span(name)
            t = arena[o:o + n]
            return t.view(shape) if shape is not None else t
        day_w = arena[self._span("day_weights.0")[0]:]
        day_b = arena[self._span("day_biases.0")[0]:]
        return dict(day_w=day_w, day_b=day_b, day_w_stride=ops.pad_to(F * F), day_b_stride=ops.pad_to(F),
                    w_ih=[v(f"gru.weight_ih_l{l}") for l in range(L)], w_hh=[v(f"gru.weight_hh_l{l}") for l in range(L)],
                    b_ih=[v(f"gru.bias_ih_l{l}") for l in range(L)], b_hh=[v(f"gru.bias_hh_l{l}") for l in range(L)],
                    out_w=v("out.weight"), out_b=v("out.bias"), h0=v("h0"))

    def _kernel_params(self) -> ops.Params:
        if not self._arena_ok():
            self.pack()
        if self._kp is None:
            self._kp = ops.Params(**self._views(self._arena))
        return self._kp

    def arena_grads(self) -> ops.Grads:
        """Gradient destinations inside the gradient arena (the trainer's fused step writes here)."""
        return ops.Grads(**self._views(self.grad_arena()))

    def _fresh_grads(self, device, day_idx):
        """Standalone gradient buffers for the autograd path (same layout as the arena)."""
        holder = torch.zeros_like(self.arena())
        return ops.Grads(**self._views(holder)), holder

    def _collect_grads(self, holder, day_idx):
        active = set(int(d) for d in day_idx.tolist())
        grads = []
        for (name, p), (o, n) in zip(self._param_order(), self.layout()["spans"]):
            if name.startswith("day_") and int(name.split(".")[1]) not in active:
                grads.append(None)     # like the reference: days absent from the batch get no gradient
            else:
                grads.append(holder[o:o + n].view(p.shape))
        return grads

    # ------------------------------------------------------------------ misc -------------------
    def _p_in(self):
        return float(self.input_dropout) if self.training else 0.0

    def _p_rnn(self):
        return float(self.rnn_dropout) if self.training else 0.0

    def _next_seed(self):
        self._seed_ctr += 1
        return (self._seed_base * 7919 + self._seed_ctr) & 0x7FFFFFFF

    def _prep_inputs(self, x, day_idx):
        if not x.is_cuda:
            raise RuntimeError("GRUDecoder.forward needs inputs on the HIP device; there is no CPU path "
                               "(the CPU restatement lives in oracle/ for tests only)")
        if x.dtype != torch.float32:
            x = x.float()      # the reference feeds bf16 under autocast (evaluate_model.py:118); math here is fp32
        x = x.contiguous()
        if not isinstance(day_idx, torch.Tensor):
            day_idx = torch.as_tensor(list(day_idx))
        day_idx = day_idx.to(device=x.device, dtype=torch.int32).contiguous().view(-1)
        if day_idx.numel() != x.shape[0]:
            raise RuntimeError("day_idx must have one entry per batch row")
        return x, day_idx

    # ------------------------------------------------------------------ streaming calls as hipGraphs -----
    # The online decoder calls forward() once per 80 ms frame with the carried states (evaluate_model_helpers.py:87-115): a dozen
    # dependent launches of a few microseconds each -- 0.17 ms of device time, 0.29 ms per call with the host's share.  The
    # call is fixed-shape and runs on one stream, so from the third call of a shape on it is replayed as ONE hipGraph
    # (torch.cuda.CUDAGraph around the same executor call: same kernels, same results bit for bit): 38 us of host time per call
    # instead of 99, 0.210 ms per call with the synchronisation instead of 0.219 (the device's 0.166 ms are the floor; the larger
    # gain of late round 3, 0.288 -> 0.219, was the parameter-arena check in front of every forward, see _arena_ok).
    # Inputs are copied into the graph's static buffers and the results into fresh tensors by kernels inside the graph (the
    # caller may keep them across calls).  The graph reads the parameter arena in place, so weight updates between calls are
    # seen; a re-packed arena, a different shape or precision mode gets a graph of its own (at most 8 are kept).
    # B2T_STREAM_GRAPH=0 turns it off.
    def _graph_eligible(self, x):
        if not ops.STREAM["graph"] or self.training or x.shape[0] > 64 or x.device.index != torch.cuda.current_device():
            return False
        Tp = self._dims.out_T(x.shape[1])
        return 0 < Tp <= 8 and not torch.cuda.is_current_stream_capturing()

    def _graph_forward(self, x, day_idx, states):
        arena = self.arena()
        if states is not None and tuple(states.shape) != (self.n_layers, x.shape[0], self.n_units):
            raise ValueError(f"states must be [{self.n_layers},{x.shape[0]},{self.n_units}], got {tuple(states.shape)}")   # (the replay copies the captured element count: never read a smaller tensor out of bounds)
        # the capture stream is part of the key: the static buffers and the `done` event are ordered on that stream only, a call
        # from another current stream gets (eager calls, then) a graph of its own
        key = (tuple(x.shape), None if states is None else tuple(states.shape), arena.data_ptr(), x.device.index, bool(ops.AMP["on"]),
               bool(ops.STREAM["fused"]), int(torch.cuda.current_stream(x.device).cuda_stream))
        ent = self._graphs.get(key)
        if ent is None:
            if len(self._graphs) >= 8:
                self._graphs.pop(next(iter(self._graphs)))
            ent = self._graphs[key] = {"calls": 0, "graph": None}
        ent["calls"] += 1
        if ent["graph"] is None:
            if ent["calls"] < 3:          # the first calls of a shape run eagerly (they are also the warm-up the capture needs)
                lg, hd, _ = ops.model_forward(self._dims, self._kernel_params(), x, day_idx, states, self._ws, save=False)
                return lg, hd
            self._capture(ent, x, day_idx, states)
        # One replay = copy this call's inputs into the static buffers, the pass, copy the results into this call's fresh output
        # tensors: the two copies are kernels INSIDE the graph that read their pointers from a pinned table when they run
        # (b2t_copy_indirect_b32), so the host's share of a call is two allocations, ten table entries and the graph launch.
        if not ent["done"].query():
            ent["done"].synchronize()     # the previous replay still reads the table (back-to-back calls without a host read)
        logits, hidden = torch.empty_like(ent["logits"]), torch.empty_like(ent["hidden"])
        ti, to = ent["tab_in"], ent["tab_out"]
        ti[1], ti[2] = x.data_ptr(), day_idx.data_ptr()
        if states is not None:
            ti[3] = states.data_ptr()
        to[5], to[6] = logits.data_ptr(), hidden.data_ptr()
        ent["graph"].replay()
        ent["done"].record()
        ent["keep"] = (x, day_idx, states)   # the replay reads them asynchronously: keep them alive until the next call
        return logits, hidden

    def _capture(self, ent, x, day_idx, states):
        import ctypes as C
        import b2t_native as N
        lib = N.load()
        sx, sd = torch.empty_like(x), torch.empty_like(day_idx)
        ss = torch.empty_like(states) if states is not None else None
        pins = torch.zeros((2, 16), dtype=torch.int64).pin_memory()
        ti, to = pins[0].numpy(), pins[1].numpy()
        ins = [(x, sx), (day_idx, sd)] + ([(states, ss)] if states is not None else [])
        ti[0] = len(ins)
        for k, (src, dst) in enumerate(ins):
            ti[1 + k], ti[5 + k], ti[9 + k] = src.data_ptr(), dst.data_ptr(), src.numel()
        blocks = 256
        stream_of = lambda: C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):      # other host threads (n-best pool, loaders) are left alone
            N.check(lib.b2t_copy_indirect_b32(C.c_void_p(pins[0].data_ptr()), blocks, stream_of()), "b2t_copy_indirect_b32")
            lg, hd, _ = ops.model_forward(self._dims, self._kernel_params(), sx, sd, ss, self._ws, save=False)
            to[0] = 2
            to[1], to[2] = lg.data_ptr(), hd.data_ptr()
            to[9], to[10] = lg.numel(), hd.numel()
            N.check(lib.b2t_copy_indirect_b32(C.c_void_p(pins[1].data_ptr()), blocks, stream_of()), "b2t_copy_indirect_b32")
        ent.update(graph=g, logits=lg, hidden=hd, x=sx, day=sd, states=ss, pins=pins, tab_in=ti, tab_out=to,
                   done=torch.cuda.Event())
        ent["done"].record()

    # ------------------------------------------------------------------ forward ----------------
    def forward(self, x, day_idx, states=None, return_state=False):
        """x [B,T,neural_dim]; day_idx [B]; states [n_layers,B,n_units] or None.
        Returns logits [B,T',n_classes] (and hidden states [n_layers,B,n_units])."""
        x, day_idx = self._prep_inputs(x, day_idx)
        if self.arena().device != x.device:
            self.pack(x.device)
        if states is not None:
            states = states.to(device=x.device, dtype=torch.float32).contiguous()
        want_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        if want_grad:
            params = [p for _, p in self._param_order()]
            logits, hidden = _ModelFn.apply(self, x, day_idx, states, True, *params)
        elif self._graph_eligible(x):
            logits, hidden = self._graph_forward(x, day_idx, states)
        else:
            logits, hidden, _ = ops.model_forward(self._dims, self._kernel_params(), x, day_idx, states, self._ws,
                                                  save=False, in_drop=self._p_in(), rnn_drop=self._p_rnn(),
                                                  seed=self._next_seed())
        if return_state:
            return logits, hidden
        return logits
