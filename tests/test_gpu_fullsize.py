"""BASELINE configs[1] at its FULL size (B=64, T=500, F=512, H=512, L=5, C=41, fp32) through properties that do not need
an oracle run of the whole batch, plus a direct oracle comparison on a slice (sentences are independent: the oracle runs
two of the 64 alone).  Everything goes through the product path (model -> b2t_ops -> C ABI)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nejm-brain-to-text_amd"))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu

B, T, F, H, L, C, D, S = 64, 500, 512, 512, 5, 41, 45, 60


@pytest.fixture(scope="module")
def c2():
    from rnn_model import GRUDecoder
    dev = torch.device("cuda:0")
    torch.manual_seed(1234)
    model = GRUDecoder(F, H, D, C, 0.0, 0.0, L, 0, 0)
    sd = {k: v.detach().numpy().copy() for k, v in model.state_dict().items()}
    model = model.to(dev)
    g = torch.Generator().manual_seed(99)
    x = torch.randn(B, T, F, generator=g) * 0.6
    day = torch.randint(0, D, (B,), generator=g).to(torch.int32)
    tgt = torch.randint(1, C, (B, S), generator=g).to(torch.int32)
    tl = torch.randint(20, S + 1, (B,), generator=g).to(torch.int32)
    nt = torch.randint(300, T + 1, (B,), generator=g).to(torch.int32)
    for b in range(B):
        tgt[b, tl[b]:] = 0
    return dict(model=model, sd=sd, x=x, day=day, tgt=tgt, tl=tl, nt=nt, dev=dev)


def _logits(c2, x, day, states=None):
    m = c2["model"].eval()
    with torch.no_grad():
        lg, hid = m(x.to(c2["dev"]), day.to(c2["dev"]), states, True)
    torch.cuda.synchronize()
    m._ws.check_sync()
    return lg, hid


def test_full_size_slice_matches_oracle(c2):
    """Two sentences of the full batch against the oracle run on those two alone (per-sentence independence)."""
    import oracle.b2t_oracle as O
    lg, _ = _logits(c2, c2["x"], c2["day"])
    pick = [3, 41]
    ref = O.model_fwd(c2["sd"], c2["x"][pick].numpy(), c2["day"][pick].numpy(), L)[0]
    np.testing.assert_allclose(lg[pick].cpu().numpy(), ref, atol=1e-4)
    # bit-exact greedy argmax wherever the oracle's top-2 margin is clear
    top2 = np.sort(ref, axis=-1)[..., -2:]
    clear = (top2[..., 1] - top2[..., 0]) > 1e-4
    np.testing.assert_array_equal(lg[pick].cpu().numpy().argmax(-1)[clear], ref.argmax(-1)[clear])


def test_full_size_batch_permutation_is_exact(c2):
    """Sentences are independent recurrences: permuting the batch permutes the logits BIT-EXACTLY (same per-element
    summation order in every tile position), and the run is deterministic."""
    lg, hid = _logits(c2, c2["x"], c2["day"])
    lg2, hid2 = _logits(c2, c2["x"], c2["day"])
    assert torch.equal(lg, lg2) and torch.equal(hid, hid2)
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(5))
    lgp, hidp = _logits(c2, c2["x"][perm], c2["day"][perm])
    assert torch.equal(lgp, lg[perm.to(lg.device)])
    assert torch.equal(hidp, hid[:, perm.to(hid.device)])


def test_full_size_causality_and_streaming(c2):
    """The GRU is causal: frames >= t0 do not touch logits < t0 (exactly); running the 500 frames as 250 + 250 with
    the carried state reproduces the single pass exactly."""
    x = c2["x"]
    lg, hid = _logits(c2, x, c2["day"])
    x2 = x.clone()
    t0 = 317
    x2[:, t0:] += 1.0
    lg2, _ = _logits(c2, x2, c2["day"])
    assert torch.equal(lg2[:, :t0], lg[:, :t0])
    assert not torch.equal(lg2[:, t0:], lg[:, t0:])
    la, ha = _logits(c2, x[:, :250], c2["day"])
    lb, hb = _logits(c2, x[:, 250:], c2["day"], states=ha)
    assert torch.equal(torch.cat([la, lb], 1), lg)
    assert torch.equal(hb, hid)


def test_full_size_train_step_properties(c2):
    """Full training step at C2: deterministic (bit-identical loss and gradients on a repeat), loss equal to the mean of
    per-sentence CTC losses that only depend on the first n_time_steps frames, gradient of the loss scaled by 2 is exactly
    2x, and the two-sentence slice of the loss matches the oracle."""
    import b2t_ops as ops
    import oracle.b2t_oracle as O
    from b2t_train_step import TrainStep
    dev = c2["dev"]
    args = dict(lr_max=1e-30, lr_min=1e-30, lr_decay_steps=10, lr_warmup_steps=0, lr_max_day=1e-30, lr_min_day=1e-30,
                lr_decay_steps_day=10, lr_warmup_steps_day=0, beta0=0.9, beta1=0.999, epsilon=0.1, weight_decay=0.0,
                weight_decay_day=0, grad_norm_clip_value=0, _debug_keep_unclipped=True)
    model = c2["model"].train()
    ts = TrainStep(model, args)
    x, day, tgt, nt, tl = c2["x"].to(dev), c2["day"], c2["tgt"], c2["nt"], c2["tl"]
    loss1, _ = ts.step(x, day, tgt, nt, tl)
    g1 = {k: v.copy() for k, v in ts.last_unclipped_grads().items()}
    loss2, _ = ts.step(x, day, tgt, nt, tl)
    g2 = ts.last_unclipped_grads()
    torch.cuda.synchronize()
    model._ws.check_sync()
    assert float(loss1) == float(loss2)
    for k in g1:
        np.testing.assert_array_equal(g1[k], g2[k], err_msg=k)
    # frames beyond n_time_steps are invisible to the loss
    lg, _ = _logits(c2, c2["x"], c2["day"])
    model.train()
    adj = ts.adjusted_lens(nt.to(dev))
    lb, dl, ldd = ops.ctc_loss(lg, tgt, adj, tl, True, 1.0 / B, model._ws)
    assert abs(float(lb.mean()) - float(loss1)) < 1e-5 * abs(float(loss1))
    lg2 = lg.clone()
    for b in range(B):
        lg2[b, int(adj[b]):] = 7.0
    lb2, dl2, _ = ops.ctc_loss(lg2, tgt, adj, tl, True, 1.0 / B, model._ws)
    assert torch.equal(lb, lb2)
    mask = torch.arange(T, device=dev)[None, :] >= adj[:, None].to(dev)
    assert float(dl2[mask].abs().max()) == 0.0
    # gradient scale: exact in powers of two
    _, dla, _ = ops.ctc_loss(lg, tgt, adj, tl, True, 1.0 / B, model._ws)
    dla = dla.clone()
    _, dlb, _ = ops.ctc_loss(lg, tgt, adj, tl, True, 2.0 / B, model._ws)
    assert torch.equal(dlb, 2.0 * dla)
    # per-sentence losses of a slice vs the oracle
    pick = [7, 58]
    ref = O.model_loss_and_grads(c2["sd"], c2["x"][pick].numpy(), c2["day"][pick].numpy(), tgt[pick].numpy(),
                                 adj[pick].cpu().numpy(), tl[pick].numpy(), L)[1]
    np.testing.assert_allclose(lb[pick].cpu().numpy(), ref, rtol=2e-5)


def _grad_check(model, sd, x, day, tgt, nt, tl, L_, ps, st, dev, tag):
    """One full training step (default execution plan of that shape) vs ONE oracle run of the whole batch: loss and every
    parameter gradient at 1e-3 of the tensor max (SURVEY 8b numerics contract)."""
    import oracle.b2t_oracle as O
    from b2t_train_step import TrainStep
    args = dict(lr_max=1e-30, lr_min=1e-30, lr_decay_steps=10, lr_warmup_steps=0, lr_max_day=1e-30, lr_min_day=1e-30,
                lr_decay_steps_day=10, lr_warmup_steps_day=0, beta0=0.9, beta1=0.999, epsilon=0.1, weight_decay=0.0,
                weight_decay_day=0, grad_norm_clip_value=0, _debug_keep_unclipped=True)
    ts = TrainStep(model.train(), args)
    loss, gn = ts.step(x.to(dev), day, tgt, nt, tl)
    got = ts.last_unclipped_grads()
    ts.check_status()
    adj = O.adjusted_lens(nt.numpy(), ps, st)
    lo, _, _, go = O.model_loss_and_grads(sd, x.numpy(), day.numpy(), tgt.numpy(), adj, tl.numpy(), L_, ps, st)
    np.testing.assert_allclose(float(loss), float(lo), rtol=2e-5)
    assert set(got) == set(go)
    worst = ("", 0.0)
    for k, ref in go.items():
        scale = max(1e-6, float(np.abs(ref).max()))
        err = float(np.abs(got[k] - ref).max()) / scale
        if err > worst[1]:
            worst = (k, err)
        assert err <= 1e-3, (tag, k, err)
    norm_o, _ = O.clip_grad_norm(go, 0)
    np.testing.assert_allclose(float(gn), float(norm_o), rtol=1e-4)
    print(f"[{tag}] loss {float(loss):.5f} (oracle {float(lo):.5f}); worst gradient error {worst[1]:.2e} of max in {worst[0]}")


def test_full_size_c2_every_gradient_matches_oracle(c2):
    """BASELINE configs[1] at full size under the plan bench.py times (6 forward / 4 backward time chunks as a task graph on four queues, up to four sweeps in
    flight): ALL parameter gradients of the 64-sentence batch against one oracle run of the same batch."""
    import b2t_ops as ops
    assert ops.time_chunks(T, B, H) == 6
    _grad_check(c2["model"], c2["sd"], c2["x"], c2["day"], c2["tgt"], c2["nt"], c2["tl"], L, 0, 0, c2["dev"], "C2")


def test_full_size_c3_shape_every_gradient_matches_oracle():
    """BASELINE configs[2]'s model shape -- H = 768, patch 14 / stride 4 (K = 7168 input projection), B = 64, T = 500 ->
    T' = 122, 45 day layers, 4 days in the batch -- forward + backward against one oracle run (dropout 0: the masks of
    the two sides cannot coincide; the dropout arithmetic has its own tests)."""
    from rnn_model import GRUDecoder
    dev = torch.device("cuda:0")
    H3, ps, st = 768, 14, 4
    torch.manual_seed(77)
    model = GRUDecoder(F, H3, D, C, 0.0, 0.0, L, ps, st)
    sd = {k: v.detach().numpy().copy() for k, v in model.state_dict().items()}
    model = model.to(dev)
    g = torch.Generator().manual_seed(78)
    x = torch.randn(B, T, F, generator=g) * 0.6
    day = torch.tensor([3, 17, 29, 44]).repeat_interleave(B // 4).to(torch.int32)
    Sm = 40
    tgt = torch.randint(1, C, (B, Sm), generator=g).to(torch.int32)
    tl = torch.randint(10, Sm + 1, (B,), generator=g).to(torch.int32)
    nt = torch.randint(420, T + 1, (B,), generator=g).to(torch.int32)
    for b in range(B):
        tgt[b, tl[b]:] = 0
    _grad_check(model, sd, x, day, tgt, nt, tl, L, ps, st, dev, "C3 shape")
