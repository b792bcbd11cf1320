"""Round 4: the forward sweep that makes the next layer's input projection itself (b2t_gru_layer_fwd_fused_f32,
csrc/gru_persistent.hip; replaces nn.GRU's W_ih product of layers >= 1, rnn_model.py:65-72,126).  Through the C ABI:
the recurrent part is BIT-IDENTICAL to the plain persistent sweep (same instruction sequence on the same operands), the
projection equals an fp64 product to fp32 roundoff, a sweep cut into chunks equals one sweep bit for bit, ragged batches
and both hand-off scopes work; and through the model: forward and every gradient with / without the fusion agree."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nejm-brain-to-text_amd"))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu

LOCAL, PARITY = 0x400, 0x800


def _dev():
    return torch.device("cuda:0")


def _setup(T, B, H, seed):
    import b2t_native as N
    dev = _dev()
    g = torch.Generator().manual_seed(seed)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
    d = dict(gi=r(T, B, 3 * H, sc=0.5), w=r(3 * H, H, sc=0.05), b=r(3 * H, sc=0.1), h0=r(B, H, sc=0.3),
             w2=r(3 * H, H, sc=0.05), b2=r(3 * H, sc=0.1))
    d["sync"] = torch.zeros(N.load().b2t_gru_sync_bytes(T) // 4 + 16, dtype=torch.int32, device=dev)
    return d


def _plain(d, T, B, H, mode, t0=0, n=None, out=None, res=None):
    import b2t_native as N, b2t_ops as ops
    lib, p = N.load(), ops._p
    n = T if n is None else n
    out = torch.zeros(T + 1, B, H, device=_dev()) if out is None else out
    res = torch.zeros(T, B, 4 * H, device=_dev()) if res is None else res
    if t0 == 0:
        out[0] = d["h0"]
    N.check(lib.b2t_gru_layer_fwd_f32(p(d["gi"][t0:]), p(d["w"]), p(d["b"]), p(out[t0]), p(out[t0 + 1:]), p(res[t0:]), None, n, B, H,
                                      mode, p(d["sync"]), ops._stream()), "plain")
    return out, res


def _fused(d, T, B, H, mode, t0=0, n=None, out=None, res=None, gi2=None):
    import b2t_native as N, b2t_ops as ops
    lib, p = N.load(), ops._p
    n = T if n is None else n
    out = torch.zeros(T + 1, B, H, device=_dev()) if out is None else out
    res = torch.zeros(T, B, 4 * H, device=_dev()) if res is None else res
    gi2 = torch.full((T, B, 3 * H), float("nan"), device=_dev()) if gi2 is None else gi2
    if t0 == 0:
        out[0] = d["h0"]
    N.check(lib.b2t_gru_layer_fwd_fused_f32(p(d["gi"][t0:]), p(d["w"]), p(d["b"]), p(out[t0]), p(out[t0 + 1:]), p(res[t0:]), None,
                                            p(d["w2"]), p(d["b2"]), p(gi2[t0:]), n, B, H, mode, p(d["sync"]), ops._stream()), "fused")
    return out, res, gi2


def _status(d):
    torch.cuda.synchronize()
    return int(d["sync"][0].item())


@pytest.mark.parametrize("local", [0, LOCAL, LOCAL | PARITY])
@pytest.mark.parametrize("T,B,H", [(40, 64, 512), (33, 20, 256), (17, 7, 128), (25, 37, 512), (9, 64, 64)])
def test_fused_sweep_recurrence_bit_identical_projection_exact(T, B, H, local):
    d = _setup(T, B, H, 7 + T + B)
    out0, res0 = _plain(d, T, B, H, 1 | local)
    assert _status(d) == 0
    out1, res1, gi2 = _fused(d, T, B, H, 1 | local)
    assert _status(d) == 0
    assert torch.equal(out0, out1), "hidden states of the fused sweep differ from the plain persistent sweep"
    assert torch.equal(res0, res1)
    ref = (out0[1:].double().reshape(T * B, H) @ d["w2"].double().t() + d["b2"].double()).reshape(T, B, 3 * H)
    err = float((gi2.double() - ref).abs().max())
    assert not torch.isnan(gi2).any(), "the fused sweep left projection rows unwritten"
    assert err < 2e-5 * max(1.0, float(ref.abs().max())), err


def test_fused_sweep_chunks_equal_one_sweep_bit_for_bit():
    T, B, H = 96, 64, 512
    d = _setup(T, B, H, 11)
    out0, res0, gi0 = _fused(d, T, B, H, 1 | LOCAL)
    assert _status(d) == 0
    out = torch.zeros(T + 1, B, H, device=_dev()); res = torch.zeros(T, B, 4 * H, device=_dev())
    gi2 = torch.full((T, B, 3 * H), float("nan"), device=_dev())
    for t0, n in ((0, 40), (40, 16), (56, 40)):
        _fused(d, T, B, H, 1 | LOCAL, t0=t0, n=n, out=out, res=res, gi2=gi2)
    assert _status(d) == 0
    assert torch.equal(out0, out) and torch.equal(res0, res)
    assert torch.equal(gi0, gi2), "the projection must not depend on how the sweep is cut into chunks"


def test_fused_sweep_rejects_what_it_cannot_do():
    import b2t_native as N, b2t_ops as ops
    lib, p = N.load(), ops._p
    T, B, H = 4, 4, 768
    z = torch.zeros(8, device=_dev())
    rc = lib.b2t_gru_layer_fwd_fused_f32(p(z), p(z), p(z), p(z), p(z), None, None, p(z), p(z), p(z), T, B, H, 1, p(z), ops._stream())
    assert rc != 0 and "512" in N.last_error()
    rc = lib.b2t_gru_layer_fwd_fused_f32(p(z), p(z), p(z), p(z), p(z), None, None, p(z), p(z), p(z), T, B, 64, 1 | 0x100, p(z), ops._stream())
    assert rc != 0   # bf16 operands: not this kernel


def _model_case(L=3, B=40, T=70, F=64, H=512, seed=3):
    from rnn_model import GRUDecoder
    torch.manual_seed(seed)
    D, C, S = 4, 41, 9
    model = GRUDecoder(F, H, D, C, 0.0, 0.0, L, 0, 0)
    g = torch.Generator().manual_seed(seed + 1)
    x = torch.randn(B, T, F, generator=g) * 0.5
    day = torch.randint(0, D, (B,), generator=g)
    tgt = torch.randint(1, C, (B, S), generator=g)
    tl = torch.randint(1, S + 1, (B,), generator=g); nt = torch.randint(40, T + 1, (B,), generator=g)
    for b in range(B):
        tgt[b, tl[b]:] = 0
    return model, x, day, tgt, nt, tl


def test_model_forward_and_gradients_with_and_without_fusion(monkeypatch):
    """The product path (model -> executor): logits, loss and every gradient with the projection inside the sweeps
    (default) against the projection GEMMs (B2T_FUSED_PROJ=0), and both against the oracle."""
    import oracle.b2t_oracle as O
    from rnn_trainer import TrainStep
    dev = _dev()
    model, x, day, tgt, nt, tl = _model_case()
    L = 3
    sd0 = {k: v.detach().numpy().copy() for k, v in model.state_dict().items()}
    lo, _, _, go = O.model_loss_and_grads(sd0, x.numpy(), day.numpy(), tgt.numpy(), nt.numpy(), tl.numpy(), L)
    args = dict(lr_max=1e-30, lr_min=1e-30, lr_decay_steps=10, lr_warmup_steps=0, lr_max_day=1e-30, lr_min_day=1e-30,
                lr_decay_steps_day=10, lr_warmup_steps_day=0, beta0=0.9, beta1=0.999, epsilon=0.1, weight_decay=0.0,
                weight_decay_day=0, grad_norm_clip_value=0, _debug_keep_unclipped=True)
    model = model.to(dev).train()
    ts = TrainStep(model, args)
    got = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("B2T_FUSED_PROJ", flag)
        loss, _ = ts.step(x.to(dev), day, tgt, nt, tl)
        model._ws.check_sync()
        got[flag] = (float(loss), ts.last_unclipped_grads())
        with torch.no_grad():
            model.eval()
            lg, hid = model(x.to(dev), day.to(dev), None, True)
            model.train()
        got[flag] += (lg.cpu().numpy(), hid.cpu().numpy())
    np.testing.assert_allclose(got["1"][0], float(lo), rtol=2e-5)
    np.testing.assert_allclose(got["1"][0], got["0"][0], rtol=2e-6)
    np.testing.assert_allclose(got["1"][2], got["0"][2], atol=3e-5)      # logits: the two summation orders of the projection
    np.testing.assert_allclose(got["1"][3], got["0"][3], atol=3e-5)
    for k, ref in go.items():
        sc = max(1e-6, float(np.abs(ref).max()))
        np.testing.assert_allclose(got["1"][1][k], ref, atol=1e-3 * sc, err_msg=k)
        np.testing.assert_allclose(got["1"][1][k], got["0"][1][k], atol=2e-4 * sc, err_msg=k)


def test_fusion_is_off_where_dropout_sits_between_the_layers(monkeypatch):
    """Training with rnn_dropout > 0: the mask sits between out[l] and the projection, so the executor keeps the GEMMs;
    the step is the same with B2T_FUSED_PROJ=0 and 1, bit for bit."""
    from rnn_model import GRUDecoder
    from rnn_trainer import TrainStep
    dev = _dev()
    torch.manual_seed(5)
    F, H, D, C, L, B, T, S = 64, 256, 3, 41, 3, 16, 48, 6
    g = torch.Generator().manual_seed(6)
    x = torch.randn(B, T, F, generator=g) * 0.5
    day = torch.randint(0, D, (B,), generator=g)
    tgt = torch.randint(1, C, (B, S), generator=g); tl = torch.full((B,), S); nt = torch.full((B,), T)
    args = dict(lr_max=1e-3, lr_min=1e-3, lr_decay_steps=10, lr_warmup_steps=0, lr_max_day=1e-3, lr_min_day=1e-3,
                lr_decay_steps_day=10, lr_warmup_steps_day=0, beta0=0.9, beta1=0.999, epsilon=0.1, weight_decay=0.0,
                weight_decay_day=0, grad_norm_clip_value=10)
    outs = []
    for flag in ("1", "0"):
        monkeypatch.setenv("B2T_FUSED_PROJ", flag)
        torch.manual_seed(7)
        model = GRUDecoder(F, H, D, C, 0.4, 0.0, L, 0, 0).to(dev).train()
        ts = TrainStep(model, args)
        for i in range(2):
            ts.step(x.to(dev), day, tgt, nt, tl)   # dropout seeds: torch.initial_seed() + a per-model counter, equal in both runs
        torch.cuda.synchronize()
        outs.append({k: v.detach().cpu().clone() for k, v in model.state_dict().items()})
    for k in outs[0]:
        assert torch.equal(outs[0][k], outs[1][k]), k
