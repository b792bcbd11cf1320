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
        if self._arena is None:
            return False
        base = self._arena.data_ptr()
        for (name, p), (off, n) in zip(self._param_order(), self._layout["spans"]):
            if p.data_ptr() != base + 4 * off or p.device != self._arena.device:
                return False
        return True

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
        else:
            logits, hidden, _ = ops.model_forward(self._dims, self._kernel_params(), x, day_idx, states, self._ws,
                                                  save=False, in_drop=self._p_in(), rnn_drop=self._p_rnn(),
                                                  seed=self._next_seed())
        if return_state:
            return logits, hidden
        return logits
