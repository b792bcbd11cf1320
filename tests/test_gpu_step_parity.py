"""GPU parity of the training step beyond the plain C2 slice (round-2 additions): patching backward, dropout,
the per-trial evaluation step, refused steps (sweep error words / non-finite norm), the data-parallel arithmetic on one
GPU, checkpoints exchanged with torch's optimizer / scheduler and with the reference trainer, every augmentation.
Tolerances as in tests/test_gpu_parity.py: gradients 1e-3 of the tensor max, parameters after AdamW steps abs 2e-5.
"""
import os
import sys
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import b2t_oracle as O


def _dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    return torch.device("cuda:0")


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def sd_of(z, prefix="sd::"):
    return {k[len(prefix):]: z[k] for k in z.files if k.startswith(prefix)}


def make_model(cfg, sd, drop=(0.0, 0.0)):
    from rnn_model import GRUDecoder
    F, H, D, C, L, ps, st = [int(v) for v in cfg]
    m = GRUDecoder(F, H, D, C, drop[0], drop[1], L, ps, st)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    return m.to(_dev())


def step_args(z=None, **kw):
    w = int(z["warmup"]) if z is not None else 4
    a = dict(lr_max=0.005, lr_min=0.0001, lr_decay_steps=120000, lr_warmup_steps=w, lr_max_day=0.005, lr_min_day=0.0001,
             lr_decay_steps_day=120000, lr_warmup_steps_day=w, beta0=0.9, beta1=0.999, epsilon=0.1, weight_decay=0.001,
             weight_decay_day=0, grad_norm_clip_value=float(z["clip"]) if z is not None else 10.0, _debug_keep_unclipped=True)
    a.update(kw)
    return a


# ------------------------------------------------------------------------------------------------
def test_train_step_patch_golden(golden_dir):
    """Four steps of the reference's step body with patch_size 14 / stride 4 (rnn_model.py:106-119, rnn_trainer.py:527-558):
    loss, every gradient (incl. b2t_patch_fold_f32 and layer 0's dW_ih over the overlapping im2col rows), clip, AdamW."""
    import b2t_ops as ops
    from b2t_train_step import TrainStep
    z = load(golden_dir, "train_step_patch.npz")
    dev = _dev()
    assert int(z["cfg"][5]) == 14 and int(z["cfg"][6]) == 4
    m = make_model(z["cfg"], sd_of(z, "sd0::")).train()
    ts = TrainStep(m, step_args(z))
    x = torch.from_numpy(z["x"]).to(dev)
    day, tgt, tl, nts = (torch.from_numpy(z[k]) for k in ("day_idx", "targets", "tgt_len", "n_time_steps"))
    gold = sd_of(z, "grad0::")
    for it in range(4):
        feats = ops.augment_smooth(x, 2, 100, "same")
        loss, gnorm = ts.step(feats, day, tgt, nts, tl)
        np.testing.assert_allclose(float(loss), z[f"loss{it}"], rtol=2e-5)
        np.testing.assert_allclose(float(gnorm), z[f"gnorm{it}"], rtol=1e-4)
        if it == 0:
            np.testing.assert_allclose(ts.last_logits.cpu().numpy(), z["logits0"], atol=1e-4)
            g = ts.last_unclipped_grads()
            assert set(g) == set(gold)
            for k, ref in gold.items():
                np.testing.assert_allclose(g[k], ref, atol=1e-3 * max(1e-6, float(np.abs(ref).max())), err_msg=k)
        sd = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
        for k, ref in sd_of(z, f"sd{it+1}::").items():
            np.testing.assert_allclose(sd[k], ref, atol=2e-5, err_msg=f"step{it} {k}")
    ts.check_status()


def test_run_single_decoding_step_golden(golden_dir):
    """evaluate_model_helpers.runSingleDecodingStep (reference :87-115): 'valid' smoothing -> patch 14/4 model in eval mode
    (dropout configured but inactive) -> fp32 numpy logits, against the reference's output on the same trial."""
    from evaluate_model_helpers import runSingleDecodingStep
    z = load(golden_dir, "evalstep.npz")
    m = make_model(z["cfg"], sd_of(z), drop=(0.4, 0.2)).eval()
    margs = dict(dataset=dict(data_transforms=dict(smooth_kernel_std=2, smooth_kernel_size=100)))
    lg = runSingleDecodingStep(torch.from_numpy(z["x"]), int(z["day"]), m, margs, "cuda:0")
    assert lg.dtype == np.float32 and lg.shape == z["logits"].shape        # T'' = floor((T - 8 - 14) / 4) + 1
    np.testing.assert_allclose(lg, z["logits"], atol=1e-4)
    srt = np.sort(z["logits"], axis=-1)
    safe = (srt[..., -1] - srt[..., -2]) > 1e-4
    assert np.array_equal(np.argmax(lg, -1)[safe], np.argmax(z["logits"], -1)[safe])
    # bf16 input, as evaluate_model.py:118 passes it: the input is rounded, the arithmetic stays fp32
    lg16 = runSingleDecodingStep(torch.from_numpy(z["x"]).to(torch.bfloat16), int(z["day"]), m, margs, "cuda:0")
    assert np.abs(lg16 - z["logits"]).max() < 0.15


# ------------------------------------------------------------------------------------------------
def test_dropout_kernels():
    """b2t_dropout_f32 / b2t_dropout_mask_f32 (rnn_model.py:102-103, nn.GRU dropout): keep rate 1-p, survivors scaled by
    1/(1-p), deterministic under the seed, a different seed gives a different mask, the gradient pass applies the SAME
    mask, a tensor processed in chunks (elem0) gets the mask of the whole tensor."""
    import b2t_native as Nn
    import b2t_ops as ops
    dev = _dev()
    lib = Nn.load()
    n = 1 << 20
    x = torch.randn(n, device=dev) + 3.0          # no zeros in the input
    for p in (0.2, 0.4):
        y = torch.empty_like(x); y2 = torch.empty_like(x); y3 = torch.empty_like(x)
        ops.dropout(x, y, n, p, seed=1234)
        ops.dropout(x, y2, n, p, seed=1234)
        ops.dropout(x, y3, n, p, seed=1235)
        assert torch.equal(y, y2) and not torch.equal(y, y3)
        kept = y != 0
        keep_rate = float(kept.float().mean())
        assert abs(keep_rate - (1 - p)) < 4 * np.sqrt(p * (1 - p) / n) + 1e-4, keep_rate
        np.testing.assert_allclose(y[kept].cpu().numpy(), (x[kept] / (1 - p)).cpu().numpy(), rtol=1e-6)
        # backward: same seed on the incoming gradient -> same zeros, same scale
        dy = torch.randn(n, device=dev) + 5.0
        dx = torch.empty_like(dy)
        ops.dropout(dy, dx, n, p, seed=1234)
        assert torch.equal(dx != 0, kept)
        np.testing.assert_allclose(dx[kept].cpu().numpy(), (dy[kept] / (1 - p)).cpu().numpy(), rtol=1e-6)
        # the factor tensor is the same mask
        mk = torch.empty_like(x)
        Nn.check(lib.b2t_dropout_mask_f32(ops._p(mk), n, float(p), 1234, 0, ops._stream()), "mask")
        assert torch.equal(mk != 0, kept) and abs(float(mk.max()) - 1 / (1 - p)) < 1e-6
        # chunked == whole
        yc = torch.empty_like(x)
        c0 = 4096 * 37
        ops.dropout(x, yc, c0, p, seed=1234)
        ops.dropout(x, yc, n - c0, p, seed=1234, elem0=c0, x_off=c0, y_off=c0)
        assert torch.equal(yc, y)
        # independence of neighbours
        k = kept.float() - (1 - p)
        assert abs(float((k[:-1] * k[1:]).mean())) < 5e-3


@pytest.mark.parametrize("patch", [(0, 0), (6, 2)])
def test_dropout_step_gradient_is_the_derivative_of_the_dropped_forward(patch):
    """Input dropout + inter-layer dropout end to end: with the masks pinned (same seed) the loss is a deterministic
    function of the parameters, and the gradient the step produces is its derivative -- checked against central
    differences along random directions (forward and backward therefore apply the same masks with the same scale)."""
    import b2t_ops as ops
    from rnn_model import GRUDecoder
    from b2t_train_step import TrainStep
    dev = _dev()
    F, H, D, C, L, B, T, S = 16, 32, 3, 41, 3, 12, 40, 4
    torch.manual_seed(2)
    model = GRUDecoder(F, H, D, C, 0.3, 0.2, L, patch[0], patch[1]).to(dev).train()
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(B, T, F, generator=g) * 0.7).to(dev)
    day = torch.randint(0, D, (B,), generator=g)
    tgt = torch.randint(1, C, (B, S), generator=g); tl = torch.full((B,), S); nt = torch.full((B,), T)
    ts = TrainStep(model, step_args(lr_max=1e-30, lr_min=1e-30, lr_max_day=1e-30, lr_min_day=1e-30, weight_decay=0.0,
                                    grad_norm_clip_value=0))
    ctr = model._seed_ctr

    def loss_at():
        model._seed_ctr = ctr            # same dropout masks every time
        lb = ts.compute_grads(x, day, tgt, nt, tl)
        return float(lb.double().mean())

    l0 = loss_at()
    grad = ts.grad_arena.clone()
    assert loss_at() == l0               # deterministic under the seed
    model._seed_ctr = ctr + 5
    assert float(ts.compute_grads(x, day, tgt, nt, tl).double().mean()) != l0   # other masks, other loss
    arena = model.arena()
    base = arena.clone()
    act = torch.zeros_like(arena)
    lay = model.layout()
    for s, (name, (o, n)) in enumerate(zip(lay["names"], lay["spans"])):
        if not name.startswith("day_") or int(name.split(".")[1]) in set(day.tolist()):
            act[o:o + n] = 1
    gen = torch.Generator(device="cpu").manual_seed(9)
    ghat = grad * act
    ghat = ghat / ghat.norm()
    for trial in range(3):
        # along the gradient itself, then two directions half gradient / half random (a purely random direction in ~20 k
        # dimensions changes the loss by less than its fp32 noise)
        d = ghat.clone()
        if trial > 0:
            r = (torch.randn(arena.shape, generator=gen).to(dev)) * act
            d = ghat + r / r.norm()
        d = d / d.norm()
        eps = 2e-2
        arena.copy_(base + eps * d); lp = loss_at()
        arena.copy_(base - eps * d); lm = loss_at()
        arena.copy_(base)
        fd = (lp - lm) / (2 * eps)
        an = float((grad.double() * d.double()).sum())
        assert abs(fd - an) <= 0.03 * max(abs(an), 0.05), (trial, fd, an)


# ------------------------------------------------------------------------------------------------
def _small_step(dev, seed=4, H=64, L=2, B=16, T=32, F=32):
    from rnn_model import GRUDecoder
    from b2t_train_step import TrainStep
    torch.manual_seed(seed)
    model = GRUDecoder(F, H, 4, 41, 0.0, 0.0, L, 0, 0).to(dev).train()
    g = torch.Generator().manual_seed(seed + 1)
    x = torch.randn(B, T, F, generator=g).to(dev)
    day = torch.randint(0, 4, (B,), generator=g)
    tgt = torch.randint(1, 41, (B, 5), generator=g); tl = torch.full((B,), 5); nt = torch.full((B,), T)
    ts = TrainStep(model, step_args(lr_warmup_steps=0, lr_warmup_steps_day=0))
    return model, ts, (x, day, tgt, nt, tl)


def test_hand_off_timeout_never_reaches_adamw():
    """A persistent sweep that reports a hand-off timeout only sets its sticky error word.  b2t_grad_norm_clip_f32 sees
    the word on the device and b2t_adamw_f32 then leaves parameters, moments and step counters untouched; the host raises
    at its next read.  Same outcome as clip_grad_norm_(error_if_nonfinite=True) at rnn_trainer.py:551-555: a bad step
    aborts, it is never applied."""
    dev = _dev()
    model, ts, batch = _small_step(dev)
    ts.step(*batch)
    ts.check_status()
    p1 = model.arena().clone(); m1 = ts.exp_avg.clone(); k1 = ts.seg_step.clone(); it1 = ts.it
    assert not torch.equal(p1, torch.zeros_like(p1)) and int(k1.max()) == 1
    sync = model._ws.sync(model.n_layers, dev)
    sync[model.n_layers + 1, 0] = 1                 # backward sweep of layer 1 "timed out"
    ts.step(*batch)
    torch.cuda.synchronize()
    assert float(ts.stat[3]) == 1.0
    assert torch.equal(model.arena(), p1) and torch.equal(ts.exp_avg, m1) and torch.equal(ts.seg_step, k1)
    with pytest.raises(RuntimeError, match="hand-off timed out"):
        ts.check_status()
    with pytest.raises(RuntimeError, match="hand-off timed out"):
        model._ws.check_sync()
    # sticky: later steps are refused too, even with the word cleared, until the owner resets the status
    sync[model.n_layers + 1, 0] = 0
    ts.step(*batch)
    assert float(ts.stat[3]) == 1.0 and torch.equal(model.arena(), p1)
    ts.stat.zero_()
    ts.step(*batch)
    ts.check_status()
    assert not torch.equal(model.arena(), p1) and int(ts.seg_step.max()) == 2 and ts.it == it1 + 3


def test_non_finite_gradient_norm_refuses_the_step():
    """clip_grad_norm_(error_if_nonfinite=True) (rnn_trainer.py:551-555): an infeasible sentence (target longer than the
    input allows) makes the CTC loss inf (zero_infinity=False, :242) and the gradients non-finite -> status 2, nothing
    applied, RuntimeError with torch's wording."""
    dev = _dev()
    model, ts, (x, day, tgt, nt, tl) = _small_step(dev)
    p0 = model.arena().clone()
    nt = nt.clone(); nt[3] = 2                       # 2 frames for 5 labels
    loss, gn = ts.step(x, day, tgt, nt, tl)
    assert not np.isfinite(float(loss))
    assert float(ts.stat[3]) == 2.0 and torch.equal(model.arena(), p0) and int(ts.seg_step.max()) == 0
    with pytest.raises(RuntimeError, match="non-finite"):
        ts.check_status()


def test_trainer_raises_on_refused_step(tmp_path):
    from test_gpu_trainer import make_args
    from rnn_trainer import BrainToTextDecoder_Trainer
    args = make_args(str(tmp_path), n_batches=8, patch=(0, 0), dropout=(0.0, 0.0))
    args['batches_per_val_step'] = 1000
    tr = BrainToTextDecoder_Trainer(args)
    tr.model._ws.sync(tr.model.n_layers, tr.device)[0, 0] = 1
    with pytest.raises(RuntimeError, match="hand-off timed out"):
        tr.train()


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("world", [2, 4])
def test_data_parallel_arithmetic_equals_one_rank(world):
    """SURVEY 8e parity mode on one GPU: the global batch split into `world` contiguous shards with different day sets,
    each shard's forward/backward scaled 1/(B_global) into its own gradient arena with the day region zeroed first, arenas
    summed (what the bucketed all-reduce does), the 'has a gradient' flags MAX-united, then clip + AdamW -- equals the
    one-rank step on the whole batch up to fp32 summation order, for the gradients, the norm and the updated parameters,
    including which day tensors are touched."""
    from rnn_model import GRUDecoder
    from b2t_train_step import TrainStep
    dev = _dev()
    F, H, D, C, L, B, T, S = 32, 64, 8, 41, 2, 16, 36, 5
    g = torch.Generator().manual_seed(11)
    x = torch.randn(B, T, F, generator=g).to(dev)
    day = torch.tensor([0, 0, 3, 3, 3, 3, 5, 5, 5, 5, 6, 6, 6, 6, 1, 1])       # shards see different days
    tgt = torch.randint(1, C, (B, S), generator=g); tl = torch.randint(2, S + 1, (B,), generator=g)
    nt = torch.randint(28, T + 1, (B,), generator=g)
    for b in range(B):
        tgt[b, tl[b]:] = 0
    args = step_args(lr_warmup_steps=0, lr_warmup_steps_day=0, grad_norm_clip_value=0.5)

    def fresh():
        torch.manual_seed(21)
        m = GRUDecoder(F, H, D, C, 0.0, 0.0, L, 0, 0)
        with torch.no_grad():
            for w in m.day_weights:
                w.add_(torch.randn(w.shape) * 0.05)
        return m.to(dev).train()

    ref_m = fresh()
    ref = TrainStep(ref_m, dict(args))
    ref.step(x, day, tgt, nt, tl)
    ref.check_status()

    n = B // world
    ranks = []
    for r in range(world):
        m = fresh()
        ts = TrainStep(m, dict(args), world=world)
        sl = slice(r * n, (r + 1) * n)
        ts.compute_grads(x[sl].contiguous(), day[sl], tgt[sl], nt[sl], tl[sl], reduce=False)
        ranks.append((m, ts))
    total = sum(ts.grad_arena for _, ts in ranks)
    active = torch.stack([ts.active for _, ts in ranks]).max(0).values
    for m, ts in ranks:
        ts.grad_arena.copy_(total); ts.active.copy_(active)
        ts.apply_update()
        ts.check_status()
    m0, ts0 = ranks[0]
    assert torch.equal(active, ref.active)
    np.testing.assert_allclose(float(ts0.stat[1]), float(ref.stat[1]), rtol=2e-5)
    gr, g0 = ref._unclipped.cpu().numpy(), ts0._unclipped.cpu().numpy()
    lay = ref_m.layout()
    for s, (name, (o, cnt)) in enumerate(zip(lay["names"], lay["spans"])):
        if int(ref.active[s]):
            np.testing.assert_allclose(g0[o:o + cnt], gr[o:o + cnt], atol=2e-5 * max(1e-6, float(np.abs(gr[o:o + cnt]).max())), err_msg=name)
    np.testing.assert_allclose(m0.arena().cpu().numpy(), ref_m.arena().cpu().numpy(), atol=2e-6)
    for m, _ in ranks[1:]:
        assert torch.equal(m.arena(), m0.arena())                 # replicas stay bit-identical
    sd0 = dict(fresh().state_dict())
    untouched = [d for d in range(D) if d not in set(day.tolist())]
    for d in untouched:
        assert torch.equal(m0.state_dict()[f"day_weights.{d}"].cpu(), sd0[f"day_weights.{d}"].cpu())


# ------------------------------------------------------------------------------------------------
def test_checkpoint_from_reference_trainer_resumes_to_the_same_step(golden_dir):
    """tests/golden/ckpt_ref.pt was written by the reference's save_model_checkpoint (rnn_trainer.py:387-406) after three
    optimizer steps (torch AdamW state, LambdaLR state).  Loaded through this trainer's checkpoint path, the 4th step
    must land on the reference's 4th-step parameters: per-parameter step counters, both moments and the schedule
    position all carried over."""
    from rnn_trainer import _OptimizerAdapter, _SchedulerAdapter, _strip_prefix
    from b2t_train_step import TrainStep
    import b2t_ops as ops
    z = load(golden_dir, "ckpt_ref_step4.npz")
    ck = torch.load(os.path.join(golden_dir, "ckpt_ref.pt"), weights_only=False, map_location="cpu")
    assert set(ck) == {'model_state_dict', 'optimizer_state_dict', 'scheduler_state_dict', 'val_PER', 'val_loss'}
    dev = _dev()
    m = make_model(z["cfg"], {k: v.numpy() for k, v in _strip_prefix(ck['model_state_dict']).items()}).train()
    ts = TrainStep(m, step_args(z))
    _OptimizerAdapter(ts).load_state_dict(ck['optimizer_state_dict'])
    _SchedulerAdapter(ts).load_state_dict(ck['scheduler_state_dict'])
    assert ts.it == 3 and int(ts.seg_step.max()) == 3
    np.testing.assert_allclose(ts.current_lrs(), z["lr3"], rtol=1e-12)
    x = torch.from_numpy(z["x"]).to(dev)
    feats = ops.augment_smooth(x, 2, 100, "same")
    loss, gn = ts.step(feats, torch.from_numpy(z["day_idx"]), torch.from_numpy(z["targets"]), torch.from_numpy(z["n_time_steps"]),
                       torch.from_numpy(z["tgt_len"]))
    np.testing.assert_allclose(float(loss), z["loss3"], rtol=2e-5)
    np.testing.assert_allclose(float(gn), z["gnorm3"], rtol=1e-4)
    sd = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    for k, ref in sd_of(z, "sd4::").items():
        np.testing.assert_allclose(sd[k], ref, atol=2e-5, err_msg=k)


def test_checkpoint_loads_into_torch_optimizer_and_scheduler(golden_dir):
    """The other direction: the optimizer / scheduler dicts this trainer saves carry torch's full key set
    (tests/golden/lr_table.npz records the key sets of real AdamW / LambdaLR / LinearLR objects) and load into
    torch.optim.AdamW + LambdaLR; a torch step taken from there equals this path's next step."""
    from b2t_train_step import TrainStep, cosine_lr_factor, param_group_of
    from rnn_trainer import _OptimizerAdapter, _SchedulerAdapter
    import b2t_ops as ops
    dev = _dev()
    z = load(golden_dir, "train_step.npz")
    zk = load(golden_dir, "lr_table.npz")
    m = make_model(z["cfg"], sd_of(z, "sd0::")).train()
    args = step_args(z); args["lr_scheduler_type"] = "cosine"
    ts = TrainStep(m, args)
    x = torch.from_numpy(z["x"]).to(dev)
    batch = (torch.from_numpy(z["day_idx"]), torch.from_numpy(z["targets"]), torch.from_numpy(z["n_time_steps"]), torch.from_numpy(z["tgt_len"]))
    feats = ops.augment_smooth(x, 2, 100, "same")
    for _ in range(2):
        ts.step(feats, *batch)
    osd, ssd = _OptimizerAdapter(ts).state_dict(), _SchedulerAdapter(ts).state_dict()
    assert set(zk["optim_group_keys"]) <= set(osd["param_groups"][0].keys())
    assert set(zk["lambda_sd_keys"]) <= set(ssd.keys())
    # a torch model with the same parameters, grouped the reference's way (rnn_trainer.py:267-269)
    import copy
    params = {k: torch.nn.Parameter(v.detach().cpu().clone()) for k, v in m.state_dict().items()}
    named = list(params.items())
    groups = [[p for n, p in named if param_group_of(n) == g] for g in (0, 1, 2)]
    opt = torch.optim.AdamW([dict(params=groups[0], weight_decay=0, group_type='bias'),
                             dict(params=groups[1], lr=0.005, weight_decay=0, group_type='day_layer'),
                             dict(params=groups[2], group_type='other')], lr=0.005, betas=(0.9, 0.999), eps=0.1, weight_decay=0.001)
    w = int(z["warmup"])
    lam = lambda s: cosine_lr_factor(s, 0.0001 / 0.005, 120000, w)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, [lam, lam, lam], -1)
    osd_cpu = copy.deepcopy(osd)
    for g in osd_cpu["param_groups"]:
        g["fused"] = None                     # (fused=True needs GPU tensors; the reference moves state to its device)
    opt.load_state_dict(osd_cpu)
    sched.load_state_dict(ssd)
    assert sched.last_epoch == 2
    np.testing.assert_allclose([g["lr"] for g in opt.param_groups], ts.current_lrs(), rtol=1e-12)
    # third step on both sides from the same gradients
    ts.step(feats, *batch)
    grads = ts.last_unclipped_grads()
    coef = float(ts.stat[2])
    for n, p in named:
        p.grad = torch.from_numpy(grads[n] * np.float32(coef)) if n in grads else None
    opt.step()
    sd = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    for n, p in named:
        np.testing.assert_allclose(sd[n], p.detach().numpy(), atol=2e-6, err_msg=n)
    # LinearLR dict
    args["lr_scheduler_type"] = "linear"; args["lr_decay_steps"] = 50
    ts2 = TrainStep(make_model(z["cfg"], sd_of(z, "sd0::")).train(), args)
    ts2.it = 7
    lsd = _SchedulerAdapter(ts2).state_dict()
    assert set(zk["linear_sd_keys"]) <= set(lsd.keys())
    lin = torch.optim.lr_scheduler.LinearLR(opt, start_factor=1.0, end_factor=0.02, total_iters=50)
    lin.load_state_dict(lsd)
    assert lin.last_epoch == 7
    np.testing.assert_allclose(ts2.current_lrs(), load(golden_dir, "lr_table.npz")["linear_lrs"][7], rtol=1e-12)


# ------------------------------------------------------------------------------------------------
def test_transform_every_augmentation_golden(golden_dir, tmp_path):
    """transform_data with static gain and random walk enabled too (rnn_trainer.py:449-453,464-465), the reference's draws
    injected, both random-walk axes; and the Philox path is deterministic under the numpy seed."""
    from test_gpu_trainer import make_args
    from rnn_trainer import BrainToTextDecoder_Trainer
    z = load(golden_dir, "transform.npz")
    dev = _dev()
    tr = BrainToTextDecoder_Trainer.__new__(BrainToTextDecoder_Trainer)
    tr.device = dev
    x = torch.from_numpy(z["x"])
    n = torch.from_numpy(z["n_time_steps"])
    real_randint = np.random.randint
    for tag, axis in (("last", -1), ("time", 1)):
        tr.transform_args = dict(white_noise_std=1.0, constant_offset_std=0.2, random_walk_std=0.05, random_walk_axis=axis,
                                 static_gain_std=0.1, random_cut=3, smooth_kernel_size=100, smooth_data=True, smooth_kernel_std=2)
        draws = dict(static_gain=torch.from_numpy(z[f"full_{tag}_sg"]).to(dev), white=torch.from_numpy(z["white"]).to(dev),
                     offset=torch.from_numpy(z["offset"].reshape(z["offset"].shape[0], -1)).to(dev),
                     random_walk=torch.from_numpy(z[f"full_{tag}_rw"]).to(dev))
        np.random.randint = lambda lo, hi=None, size=None, **kw: (1 if size is None else np.ones(size, dtype=np.int64))
        try:
            y, n2 = tr.transform_data(x, n, 'train', _draws=draws)
        finally:
            np.random.randint = real_randint
        np.testing.assert_allclose(y.cpu().numpy(), z[f"full_{tag}"], atol=5e-6)
        np.testing.assert_array_equal(n2.numpy(), z["n_time_steps"] - 1)
    np.random.seed(5); a, _ = tr.transform_data(x, n, 'train')
    np.random.seed(5); b, _ = tr.transform_data(x, n, 'train')
    np.random.seed(6); c, _ = tr.transform_data(x, n, 'train')
    assert torch.equal(a, b) and (a.shape != c.shape or not torch.equal(a, c))


# ------------------------------------------------------------------------------------------------
class _FakeH5:
    """In-memory stand-in for the h5py calls dataset.py makes (File as a context manager, group[name][:], .attrs):
    exercises the HDF5 code path of BrainToTextDataset / train_test_split_indicies without the library."""
    store = {}

    class _DS:
        def __init__(self, a):
            self.a = a

        def __getitem__(self, k):
            return self.a[k]

    class _Group(dict):
        attrs = None

    class File:
        def __init__(self, path, mode='r'):
            self.g = _FakeH5.store[path]

        def __enter__(self):
            return self.g

        def __exit__(self, *a):
            return False


def _fake_sessions(tmp, n_days=3, n_trials=(7, 5, 9), F=20):
    rng = np.random.default_rng(0)
    paths = []
    for d in range(n_days):
        p = os.path.join(tmp, f"t15.2023.08.{11 + d}", "data_train.hdf5")
        os.makedirs(os.path.dirname(p), exist_ok=True)
        open(p, "w").close()                                   # train_test_split_indicies checks os.path.exists
        f = {}
        for t in range(n_trials[d]):
            T, S = int(rng.integers(30, 60)), int(rng.integers(2, 7))
            g = _FakeH5._Group(input_features=_FakeH5._DS(rng.standard_normal((T, F)).astype(np.float32)),
                               seq_class_ids=_FakeH5._DS(np.concatenate([rng.integers(1, 41, S), np.zeros(3, int)])),
                               transcription=_FakeH5._DS(np.arange(8)))
            g.attrs = dict(n_time_steps=T, seq_len=S, block_num=d + 1, trial_num=t)
            f[f"trial_{t:04d}"] = g
        _FakeH5.store[p] = f
        paths.append(p)
    return paths


def test_resident_dataset_from_hdf5_sessions_stores_each_trial_once(tmp_path, monkeypatch):
    """The HDF5 path of dataset.py (model_training/dataset.py:100-159 layout) through an in-memory h5py stand-in:
    ResidentDataset.from_dataset keeps each unique (day, trial) ONCE however many batches sample it, and replays the
    source's batches exactly (features, labels, lengths, days, block / trial numbers)."""
    import dataset as ds
    fake = types.ModuleType("h5py"); fake.File = _FakeH5.File
    monkeypatch.setitem(sys.modules, "h5py", fake)
    paths = _fake_sessions(str(tmp_path))
    train_trials, _ = ds.train_test_split_indicies(paths, test_percentage=0, seed=1)
    assert [len(v['trials']) for v in train_trials.values()] == [7, 5, 9]
    src = ds.BrainToTextDataset(train_trials, n_batches=30, split='train', batch_size=8, days_per_batch=2, random_seed=3)
    rd = ds.ResidentDataset.from_dataset(src, device='cuda:0')
    assert rd.n_trials == 21 and len(rd) == 30                 # 240 sampled rows, 21 stored trials
    for i in (0, 7, 29):
        a, b = src[i], rd.batch_of(i)
        for k in ('input_features', 'seq_class_ids', 'n_time_steps', 'phone_seq_lens', 'day_indicies', 'block_nums', 'trial_nums'):
            np.testing.assert_array_equal(b[k].cpu().numpy(), np.asarray(a[k]), err_msg=f"{k} batch {i}")
    test_src = ds.BrainToTextDataset(train_trials, n_batches=None, split='test', batch_size=4)
    rdt = ds.ResidentDataset.from_dataset(test_src, device='cuda:0')
    assert rdt.n_trials == 21 and len(rdt) == len(test_src)
    np.testing.assert_array_equal(rdt.batch_of(1)['input_features'].cpu().numpy(), test_src[1]['input_features'].numpy())
    # replaying a large sampled source is refused before anything is copied
    monkeypatch.setattr(ds.ResidentDataset, "MAX_BYTES", 1000)
    with pytest.raises(RuntimeError, match="exceed"):
        ds.ResidentDataset.from_batches(ds.SyntheticTrials(50, 8, 3, 20, 41, 2), device='cuda:0')


# ------------------------------------------------------------------------------------------------
def test_schedule_independent(monkeypatch):
    """The pass is a task graph list-scheduled onto the caller's stream + worker queues (csrc/exec.cpp): which queue a
    task lands on must not change a single bit.  Same model, batch, chunking: 1, 2 and 3 workers, sweeps on any queue or
    on the workers only, weight gradients per chunk -> identical gradient arenas and logits (every accumulation order is a
    dependency edge of the graph, not a property of the schedule)."""
    import b2t_ops as ops
    from rnn_model import GRUDecoder
    from b2t_train_step import TrainStep
    dev = _dev()
    F, H, D, C, L, B, T, S = 64, 128, 6, 41, 3, 32, 160, 12
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, T, F, generator=g).to(dev)
    day = torch.randint(0, D, (B,), generator=g)
    tgt = torch.randint(1, C, (B, S), generator=g); tl = torch.randint(3, S + 1, (B,), generator=g)
    nt = torch.randint(120, T + 1, (B,), generator=g)
    for b in range(B):
        tgt[b, tl[b]:] = 0
    monkeypatch.setitem(ops.PIPELINE, "chunks", 5)
    monkeypatch.setitem(ops.PIPELINE, "chunks_bwd", 3)

    def grads(workers, workers_only, mask):
        monkeypatch.setenv("B2T_WORKERS", str(workers))
        if workers_only:
            monkeypatch.setenv("B2T_SWEEP_WORKERS_ONLY", "1")
        else:
            monkeypatch.delenv("B2T_SWEEP_WORKERS_ONLY", raising=False)
        monkeypatch.setitem(ops.PIPELINE, "wgrad_chunk_mask", mask)
        torch.manual_seed(3)
        m = GRUDecoder(F, H, D, C, 0.0, 0.0, L, 0, 0).to(dev).train()     # a fresh model = a fresh executor
        ts = TrainStep(m, step_args())
        loss_b = ts.compute_grads(x, day, tgt, nt, tl)
        torch.cuda.synchronize()
        return ts.grad_arena.clone(), loss_b.clone()

    for mask in (0, 0b111):
        ref, loss = grads(3, False, mask)
        assert torch.isfinite(ref).all() and float(ref.abs().max()) > 0
        for workers, only in ((1, False), (2, False), (3, True), (5, False)):
            got, l2 = grads(workers, only, mask)
            assert torch.equal(got, ref), f"workers={workers} workers_only={only} mask={mask}: gradients differ"
            assert torch.equal(l2, loss)


def test_plan_results_do_not_depend_on_task_timing(monkeypatch):
    """Round-5 verdict 1c: the hipGraph replay of the plan once ended 3e-6 off the eager plan's loss, i.e. either a runtime bug or
    an ordering the eager plan gets only by TIMING (a missing edge that a handful of placements cannot see).  B2T_EXEC_JITTER=seed
    makes csrc/exec.cpp enqueue a 10-400 us spin kernel in front of a random third of the tasks (on the task's queue, behind its
    waits): 40 seeds x (forward + backward) at the schedule-independence shape, gradients / loss / logits must equal the
    unjittered plan's bit for bit.  (tools/r5_jitter.py runs 200 seeds at this shape and at BASELINE configs[1].)"""
    import b2t_ops as ops
    from rnn_model import GRUDecoder
    from b2t_train_step import TrainStep
    dev = _dev()
    F, H, D, C, L, B, T, S = 64, 128, 6, 41, 3, 32, 160, 12
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, T, F, generator=g).to(dev)
    day = torch.randint(0, D, (B,), generator=g)
    tgt = torch.randint(1, C, (B, S), generator=g); tl = torch.randint(3, S + 1, (B,), generator=g)
    nt = torch.randint(120, T + 1, (B,), generator=g)
    for b in range(B):
        tgt[b, tl[b]:] = 0
    monkeypatch.setitem(ops.PIPELINE, "chunks", 5)
    monkeypatch.setitem(ops.PIPELINE, "chunks_bwd", 3)
    for mask in (0, 0b111):
        monkeypatch.setitem(ops.PIPELINE, "wgrad_chunk_mask", mask)
        monkeypatch.delenv("B2T_EXEC_JITTER", raising=False)
        torch.manual_seed(3)
        m = GRUDecoder(F, H, D, C, 0.0, 0.0, L, 0, 0).to(dev).train()
        ts = TrainStep(m, step_args())
        loss = ts.compute_grads(x, day, tgt, nt, tl).clone()
        torch.cuda.synchronize()
        ref = ts.grad_arena.clone()
        assert torch.isfinite(ref).all() and float(ref.abs().max()) > 0
        for seed in range(20):
            monkeypatch.setenv("B2T_EXEC_JITTER", str(seed + 100 * mask))
            ts.grad_arena.zero_()
            l2 = ts.compute_grads(x, day, tgt, nt, tl)
            torch.cuda.synchronize()
            assert torch.equal(l2, loss), f"jitter seed {seed}, mask {mask}: loss differs"
            assert torch.equal(ts.grad_arena, ref), f"jitter seed {seed}, mask {mask}: gradients differ"
        ts.check_status()


def test_xcd_local_handoff_is_bit_identical(monkeypatch):
    """The XCD-local hand-off (row groups pinned to one XCD, counters / tiles through that XCD's L2) changes where the data
    travels, not the arithmetic: gradients and loss equal the device-scope hand-off's bit for bit, for every combination
    of directions, at a shape with four row groups and one with a ragged last group."""
    import b2t_ops as ops
    from rnn_model import GRUDecoder
    from b2t_train_step import TrainStep
    dev = _dev()
    for (F, H, D, C, L, B, T, S) in ((64, 512, 4, 41, 3, 64, 96, 10), (32, 96, 3, 41, 2, 37, 50, 6)):
        g = torch.Generator().manual_seed(B)
        x = torch.randn(B, T, F, generator=g).to(dev)
        day = torch.randint(0, D, (B,), generator=g)
        tgt = torch.randint(1, C, (B, S), generator=g); tl = torch.randint(2, S + 1, (B,), generator=g)
        nt = torch.randint(T - 10, T + 1, (B,), generator=g)
        for b in range(B):
            tgt[b, tl[b]:] = 0

        def grads(dirs):
            monkeypatch.setitem(ops.LOCAL_F32, "dirs", dirs)
            torch.manual_seed(3)
            m = GRUDecoder(F, H, D, C, 0.0, 0.0, L, 0, 0).to(dev).train()
            ts = TrainStep(m, step_args())
            loss_b = ts.compute_grads(x, day, tgt, nt, tl)
            torch.cuda.synchronize()
            m._ws.check_sync()
            return ts.grad_arena.clone(), loss_b.clone()

        ref, loss = grads("")
        assert torch.isfinite(ref).all() and float(ref.abs().max()) > 0
        for dirs in ("f", "b", "fb"):
            got, l2 = grads(dirs)
            assert torch.equal(got, ref) and torch.equal(l2, loss), f"B2T_GRU_LOCAL={dirs!r} at H={H}, B={B}"


def test_bf16_mode_prepacked_weights_and_dropout_in_the_pack_are_bit_identical(monkeypatch):
    """Round 5, bf16 mode: W_ih of every layer is packed to bf16 ONCE per pass (it is the B operand of one GEMM per time chunk) and the
    inter-layer dropout of nn.GRU rides in the A pack of the next layer's projection (which also writes the dropped fp32 values the
    backward pass reads) -- csrc/exec.cpp `wpack` tasks, gemm_bf16p_pack / gemm_bf16p_run; the B operands of the whole-sequence weight
    gradients (h_{t-1}^T, x^T) are packed when the backward pass starts (`xpack`), and dW_hh / dW_ih run as two tasks.  Same values,
    same rounding, same Philox draws: logits, per-sentence losses and the whole gradient arena equal B2T_PREPACK=0 (pack per GEMM,
    separate dropout kernel) bit for bit, with and without dropout, pipelined and serial plans, patch input."""
    import b2t_ops as ops
    from rnn_model import GRUDecoder
    from b2t_train_step import TrainStep
    dev = _dev()
    old_amp = ops.AMP["on"]
    try:
        ops.set_amp(True)
        for (F, H, D, C, L, B, T, S, patch, drop, chunks) in ((64, 768, 4, 41, 3, 64, 100, 10, (4, 2), (0.4, 0.2), (3, 2)),
                                                             (64, 512, 4, 41, 3, 64, 96, 10, (0, 0), (0.0, 0.0), (3, 2)),
                                                             (64, 256, 3, 41, 2, 64, 64, 6, (0, 0), (0.3, 0.0), (1, 1))):
            g = torch.Generator().manual_seed(B + H)
            x = torch.randn(B, T, F, generator=g).to(dev)
            day = torch.randint(0, D, (B,), generator=g)
            tgt = torch.randint(1, C, (B, S), generator=g); tl = torch.randint(2, S + 1, (B,), generator=g)
            nt = torch.randint(T - 10, T + 1, (B,), generator=g)
            for b in range(B):
                tgt[b, tl[b]:] = 0
            monkeypatch.setitem(ops.PIPELINE, "chunks", chunks[0])
            monkeypatch.setitem(ops.PIPELINE, "chunks_bwd", chunks[1])

            def grads(prepack, split=True):
                monkeypatch.setenv("B2T_PREPACK", "1" if prepack else "0")
                monkeypatch.setenv("B2T_WGRAD_SPLIT", "1" if split else "0")
                torch.manual_seed(3)
                m = GRUDecoder(F, H, D, C, drop[0], drop[1], L, patch[0], patch[1]).to(dev).train()
                ts = TrainStep(m, step_args())
                loss_b = ts.compute_grads(x, day, tgt, nt, tl)
                torch.cuda.synchronize()
                ts.check_status()
                return ts.grad_arena.clone(), loss_b.clone(), ts.last_logits.clone()

            ref = grads(False)
            assert torch.isfinite(ref[0]).all() and float(ref[0].abs().max()) > 0
            for split in (True, False):      # dW_hh and dW_ih as two tasks with buffers of their own, or one after the other
                got = grads(True, split)
                for a, r, name in zip(got, ref, ("gradients", "losses", "logits")):
                    assert torch.equal(a, r), f"H={H} drop={drop} chunks={chunks} split={split}: {name} differ with pre-packed operands"
    finally:
        ops.set_amp(old_amp)
        monkeypatch.delenv("B2T_PREPACK", raising=False)
        monkeypatch.delenv("B2T_WGRAD_SPLIT", raising=False)


def test_paired_backward_sweeps_in_the_step(monkeypatch):
    """Round 5: the step with its backward sweeps as paired sweeps (W_hh^T in LDS, one per XCD set in flight, csrc/exec.cpp classes
    2..5).  The eight-way split of the contraction sums in another order than the register-resident sweep's four-way split, so
    the contract against the default path is a tolerance (2e-5 of the largest gradient); against ITSELF the path is exact: the
    same gradient arena, bit for bit, under four timing-jitter seeds (B2T_EXEC_JITTER) of the pipelined plan."""
    import b2t_ops as ops
    from rnn_model import GRUDecoder
    from b2t_train_step import TrainStep
    dev = _dev()
    for (F, H, D, C, L, B, T, S, chunks) in ((64, 512, 4, 41, 5, 64, 120, 10, (6, 4)), (32, 96, 3, 41, 2, 37, 50, 6, (2, 2)), (64, 256, 4, 41, 3, 48, 96, 8, (3, 3))):
        g = torch.Generator().manual_seed(B)
        x = torch.randn(B, T, F, generator=g).to(dev)
        day = torch.randint(0, D, (B,), generator=g)
        tgt = torch.randint(1, C, (B, S), generator=g); tl = torch.randint(2, S + 1, (B,), generator=g)
        nt = torch.randint(T - 10, T + 1, (B,), generator=g)
        for b in range(B):
            tgt[b, tl[b]:] = 0
        monkeypatch.setitem(ops.PIPELINE, "chunks", chunks[0])
        monkeypatch.setitem(ops.PIPELINE, "chunks_bwd", chunks[1])

        def grads(paired, jitter=None):
            monkeypatch.setitem(ops.PAIRED_BWD, "on", paired)
            if jitter is None:
                monkeypatch.delenv("B2T_EXEC_JITTER", raising=False)
            else:
                monkeypatch.setenv("B2T_EXEC_JITTER", str(jitter))
            torch.manual_seed(3)
            m = GRUDecoder(F, H, D, C, 0.0, 0.0, L, 0, 0).to(dev).train()
            ts = TrainStep(m, step_args())
            loss_b = ts.compute_grads(x, day, tgt, nt, tl)
            torch.cuda.synchronize()
            m._ws.check_sync()
            ts.check_status()
            return ts.grad_arena.clone(), loss_b.clone()

        ref, loss = grads(False)
        got, l2 = grads(True)
        assert torch.equal(l2, loss)                       # the forward pass is the same code
        assert not torch.equal(got, ref), "the paired sweeps did not engage"
        worst = float((got - ref).abs().max()) / float(ref.abs().max())
        assert worst < 2e-5, f"H={H} B={B}: paired vs register-resident backward sweeps differ by {worst:.2e} of the largest gradient"
        for seed in range(4):
            again, l3 = grads(True, jitter=seed)
            assert torch.equal(again, got) and torch.equal(l3, loss), f"H={H} B={B}: paired sweeps, jitter seed {seed}"
    monkeypatch.delenv("B2T_EXEC_JITTER", raising=False)


@pytest.mark.parametrize("tag", ["patch", "h256"])
def test_bf16_mode_forward_against_the_reference_autocast_forward(golden_dir, tag):
    """`use_amp: true` (rnn_args.yaml:19; rnn_trainer.py:527, :704 wrap the model call in torch.autocast(dtype=bfloat16)).  Fixture:
    the reference's GRUDecoder.forward under torch.autocast on the CPU backend, next to its fp32 logits, captured by
    tests/golden/make_golden.py (make_autocast_forward).  The two regimes do not round at the same places -- CPU autocast runs the
    day-layer einsum and the output Linear in bf16 and keeps torch's CPU GRU in fp32 (cuDNN's GRU is bf16 under CUDA autocast); this
    repository's bf16 mode rounds the operands of EVERY matrix product, recurrent ones included, and keeps fp32 accumulators,
    gates, states and outputs -- so the contract is a distance, stated here: the bf16 mode's logits are as close to the
    reference's autocast logits as those are to the reference's own fp32 logits (max |difference| <= 1.5 x that distance,
    which is ~0.6 % of the largest logit), the fp32 mode reproduces the fp32 logits to 1e-4, and wherever the fp32 margin
    between the two best phonemes exceeds that distance all three argmaxes agree."""
    import b2t_ops as ops
    z = load(golden_dir, "fwd_autocast.npz")
    dev = _dev()
    m = make_model(z[f"{tag}::cfg"], sd_of(z, f"{tag}::sd::")).eval()
    x = torch.from_numpy(z[f"{tag}::x"]).to(dev); day = torch.from_numpy(z[f"{tag}::day_idx"]).to(dev)
    ref32, ref16 = z[f"{tag}::logits_fp32"], z[f"{tag}::logits_autocast"]
    dist = float(np.abs(ref16 - ref32).max())
    old = ops.AMP["on"]
    try:
        ops.set_amp(False)
        with torch.no_grad():
            l32 = m(x, day).cpu().numpy()
        ops.set_amp(True)
        with torch.no_grad():
            l16 = m(x, day).cpu().numpy()
    finally:
        ops.set_amp(old)
    np.testing.assert_allclose(l32, ref32, atol=1e-4)
    assert 0.0 < float(np.abs(l16 - l32).max()), "the bf16 mode did not engage"
    d_ref = float(np.abs(l16 - ref16).max())
    print(f"[{tag}] reference autocast vs fp32 {dist:.4f}; bf16 mode vs reference autocast {d_ref:.4f}, vs fp32 {float(np.abs(l16 - ref32).max()):.4f}; "
          f"max |logit| {float(np.abs(ref32).max()):.3f}")
    assert d_ref <= 1.5 * dist
    top2 = np.sort(ref32, axis=-1)[..., -2:]
    clear = (top2[..., 1] - top2[..., 0]) > dist
    assert clear.mean() > 0.5
    assert np.array_equal(l16.argmax(-1)[clear], ref32.argmax(-1)[clear]) and np.array_equal(ref16.argmax(-1)[clear], ref32.argmax(-1)[clear])


def test_bf16_mode_256_tile_gemms_in_the_step_are_bit_identical(monkeypatch):
    """Round 5, bf16 mode: products whose 256 x 256 tiles fill the chip run on gemm_bf16p_kernel256 (csrc/gemm_bf16p.hip) -- in the step
    these are layer 0's weight / input gradients on PRE-PACKED operands (csrc/exec.cpp `wpack` / `xpack`: packed matrices padded to 128
    rows, so a 256-row tile can end past them: clamped loads).  Every output element sums the same products in the same order as on
    128 x 128 tiles: gradients, losses and logits of a pass with the kernel forced from 64 tiles on (B2T_GEMM_256=2; a wide layer 0:
    5632 input features) equal those with it disabled, bit for bit."""
    import b2t_ops as ops
    from rnn_model import GRUDecoder
    from b2t_train_step import TrainStep
    dev = _dev()
    old_amp = ops.AMP["on"]
    try:
        ops.set_amp(True)
        F, H, D, C, L, B, T, S = 5632, 256, 3, 41, 2, 32, 168, 8       # 3H = 768 (3 tile rows; 6 packed 128-row panels), B*T = 5376 = 21 tile rows
        g = torch.Generator().manual_seed(77)
        x = torch.randn(B, T, F, generator=g).to(dev)
        day = torch.randint(0, D, (B,), generator=g)
        tgt = torch.randint(1, C, (B, S), generator=g); tl = torch.randint(2, S + 1, (B,), generator=g)
        nt = torch.randint(T - 10, T + 1, (B,), generator=g)
        for b in range(B):
            tgt[b, tl[b]:] = 0
        monkeypatch.setitem(ops.PIPELINE, "chunks", 3)
        monkeypatch.setitem(ops.PIPELINE, "chunks_bwd", 2)

        def grads(mode):
            monkeypatch.setenv("B2T_GEMM_256", mode)
            torch.manual_seed(3)
            m = GRUDecoder(F, H, D, C, 0.2, 0.1, L, 0, 0).to(dev).train()
            ts = TrainStep(m, step_args())
            loss_b = ts.compute_grads(x, day, tgt, nt, tl)
            torch.cuda.synchronize()
            ts.check_status()
            return ts.grad_arena.clone(), loss_b.clone(), ts.last_logits.clone()

        ref = grads("0")
        assert torch.isfinite(ref[0]).all() and float(ref[0].abs().max()) > 0
        for a, r, name in zip(grads("2"), ref, ("gradients", "losses", "logits")):
            assert torch.equal(a, r), f"{name} differ between the 256-tile and the 128-tile kernel"
    finally:
        ops.set_amp(old_amp)
        monkeypatch.delenv("B2T_GEMM_256", raising=False)


def test_bf16_mode_day_layer_on_the_packed_kernel(monkeypatch):
    """Round 5, bf16 mode: the day layer's per-sentence products (forward x[b] W[day[b]] with the Softsign epilogue, backward
    x[b]^T dpre[b]; Z = B) run through the two-pass packed kernel with every matrix of the batch packed (gemm_bf16p_run, Z > 1;
    the backward's A operand -- the INPUT, transposed -- is packed when the backward pass starts: `xpack` task, csrc/exec.cpp).
    Same operands, same bf16 rounding, fp32 accumulation in another order than the one-pass kernel's: logits, losses and every
    gradient agree with B2T_ZPACK=0 to fp32 summation roundoff -- a patch model with input dropout (the shipped form) and a plain one."""
    import b2t_ops as ops
    from rnn_model import GRUDecoder
    from b2t_train_step import TrainStep
    dev = _dev()
    old_amp = ops.AMP["on"]
    try:
        ops.set_amp(True)
        for (F, H, D, C, L, B, T, S, patch, drop, chunks) in ((128, 256, 4, 41, 2, 48, 230, 8, (6, 3), (0.3, 0.2), (3, 2)),
                                                             (256, 256, 3, 41, 2, 32, 200, 8, (0, 0), (0.0, 0.0), (1, 1))):
            g = torch.Generator().manual_seed(B + H + T)
            x = torch.randn(B, T, F, generator=g).to(dev)
            day = torch.randint(0, D, (B,), generator=g)
            Tp = T if patch[0] == 0 else (T - patch[0]) // patch[1] + 1
            tgt = torch.randint(1, C, (B, S), generator=g); tl = torch.randint(2, S + 1, (B,), generator=g)
            nt = torch.randint(T - 10, T + 1, (B,), generator=g)
            for b in range(B):
                tgt[b, tl[b]:] = 0
            monkeypatch.setitem(ops.PIPELINE, "chunks", chunks[0])
            monkeypatch.setitem(ops.PIPELINE, "chunks_bwd", chunks[1])

            def grads(zpack):
                monkeypatch.setenv("B2T_ZPACK", "1" if zpack else "0")
                torch.manual_seed(3)
                m = GRUDecoder(F, H, D, C, drop[0], drop[1], L, patch[0], patch[1]).to(dev).train()
                ts = TrainStep(m, step_args())
                loss_b = ts.compute_grads(x, day, tgt, nt, tl)
                torch.cuda.synchronize()
                ts.check_status()
                return ts.grad_arena.clone(), loss_b.clone(), ts.last_logits.clone()

            ref, got = grads(False), grads(True)
            assert torch.isfinite(ref[0]).all() and float(ref[0].abs().max()) > 0
            for a, r, name in zip(got, ref, ("gradients", "losses", "logits")):
                scale = float(r.abs().max())
                assert float((a - r).abs().max()) <= 2e-4 * scale, f"patch={patch}: {name} differ by {float((a - r).abs().max())} (scale {scale})"
    finally:
        ops.set_amp(old_amp)
        monkeypatch.delenv("B2T_ZPACK", raising=False)
