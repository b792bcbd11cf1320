#!/usr/bin/env python3
"""Generate golden vectors by IMPORTING the reference (build container only).

Runs only where /root/reference exists (it does not exist on the GPU box; nothing under
tests/ reads /root/reference at test time).  The .npz files written next to this script
are data: seeded inputs + the outputs the reference modules produce for them under
torch CPU fp32.  Re-run: `python tests/golden/make_golden.py`.

Reference entry points exercised:
  model_training/rnn_model.py:GRUDecoder.forward            (fwd_*.npz, stream_*.npz)
  model_training/data_augmentations.py:gauss_smooth         (smooth.npz)
  torch.nn.CTCLoss at model_training/rnn_trainer.py:242     (ctc.npz)
  rnn_trainer.py:527-558 step body (loss/backward/clip/AdamW/LambdaLR)  (train_step.npz)
  rnn_trainer.py:294-363 create_cosine_lr_scheduler         (lr_table.npz)
  rnn_trainer.py:436-484 transform_data (torch.randn monkey-patched)    (transform.npz)
  rnn_trainer.py:724-736 / evaluate_model.py:129-141 greedy decode      (greedy.npz)
  evaluate_model_helpers.py:79-83,87-115 rearrange / runSingleDecodingStep (evalstep.npz)
  rnn_trainer.py:527-558 with patch_size 14 / stride 4                   (train_step_patch.npz)
  rnn_trainer.py:365-406 save_model_checkpoint after 3 steps + the 4th step (ckpt_ref.pt, ckpt_ref_step4.npz)
  dataset.py:162-242 create_batch_index_train / _test                    (sampler_index.npz)
  dataset.py:100-159 BrainToTextDataset.__getitem__ on an in-memory h5py   (dataset_batches.npz)
  rnn_trainer.py:228-234 LinearLR                                        (lr_table.npz: linear_*)
  rnn_trainer.py:449-465 static gain + random walk with injected draws   (transform.npz: full_*)
"""
import os
import sys
import types
import math

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REF, "model_training"))
sys.path.insert(0, REF)

# ---- stubs for packages the image lacks (h5py, torchaudio, omegaconf, redis, ...) ------------
def _stub(name, **attrs):
    import importlib.machinery
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def _edit_distance(a, b):
    a = list(a); b = list(b)
    prev = list(range(len(b) + 1))
    for i in range(1, len(a) + 1):
        cur = [i] + [0] * len(b)
        for j in range(1, len(b) + 1):
            cur[j] = min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (a[i - 1] != b[j - 1]))
        prev = cur
    return prev[len(b)]


# h5py: an in-memory File so that the reference's BrainToTextDataset.__getitem__ (dataset.py:100-159) can run without the
# package or the Dryad files.  MEMFILES[path][f"trial_{t:04d}"] = {"input_features": ..., "seq_class_ids": ...,
# "transcription": ..., "attrs": {...}}; only what __getitem__ touches is modelled (File as a context manager, group
# lookup raising KeyError, dataset[:] and group.attrs[...]).
MEMFILES = {}


class _MemGroup:
    def __init__(self, d):
        self._d = d
        self.attrs = d["attrs"]

    def __getitem__(self, k):
        return self._d[k]


class _MemFile:
    def __init__(self, path, mode="r"):
        self._f = MEMFILES[path]

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def __getitem__(self, k):
        return _MemGroup(self._f[k])


_stub("h5py", File=_MemFile)
ta = _stub("torchaudio")
taf = _stub("torchaudio.functional", edit_distance=_edit_distance)
ta.functional = taf
_stub("omegaconf", OmegaConf=type("OmegaConf", (), {"save": staticmethod(lambda *a, **k: None)}))
_stub("redis")
_stub("editdistance", eval=_edit_distance)
_stub("g2p_en", G2p=object)

from rnn_model import GRUDecoder            # noqa: E402
from data_augmentations import gauss_smooth  # noqa: E402
import rnn_trainer as ref_trainer            # noqa: E402
import evaluate_model_helpers as ref_helpers  # noqa: E402

torch.set_num_threads(8)


def sd_np(model):
    return {k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}


def perturb_days(model, gen, scale=0.05):
    with torch.no_grad():
        for w in model.day_weights:
            w.add_(torch.randn(w.shape, generator=gen) * scale)
        for b in model.day_biases:
            b.add_(torch.randn(b.shape, generator=gen) * scale)
        for n, p in model.gru.named_parameters():
            if "bias" in n:
                p.add_(torch.randn(p.shape, generator=gen) * 0.05)


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrs)
    print(f"wrote {name}: {os.path.getsize(path)/1e6:.2f} MB")


# ---------------------------------------------------------------------------------------------
def make_forward():
    cases = [
        dict(tag="h64", F=64, H=64, D=6, L=5, B=4, T=37, ps=0, st=0),
        dict(tag="h512", F=64, H=512, D=3, L=1, B=4, T=23, ps=0, st=0),
        dict(tag="patch", F=48, H=96, D=5, L=5, B=3, T=61, ps=14, st=4),
        dict(tag="f512", F=512, H=64, D=2, L=2, B=2, T=19, ps=0, st=0),
    ]
    for c in cases:
        torch.manual_seed(10)
        m = GRUDecoder(c["F"], c["H"], c["D"], 41, 0.0, 0.0, c["L"], c["ps"], c["st"]).eval()
        gen = torch.Generator().manual_seed(123)
        perturb_days(m, gen)
        x = torch.randn(c["B"], c["T"], c["F"], generator=gen)
        day = torch.randint(0, c["D"], (c["B"],), generator=gen)
        with torch.no_grad():
            logits, hs = m(x, day, None, True)
            # streaming equivalence: two chunks with carried state (patch 0 only)
            extra = {}
            if c["ps"] == 0:
                t1 = c["T"] // 2
                l1, s1 = m(x[:, :t1], day, None, True)
                l2, s2 = m(x[:, t1:], day, s1, True)
                extra = dict(stream_split=np.int64(t1), stream_logits=torch.cat([l1, l2], 1).numpy(),
                             stream_state=s2.numpy())
        arrs = {f"sd::{k}": v for k, v in sd_np(m).items()}
        save(f"fwd_{c['tag']}.npz", x=x.numpy(), day_idx=day.numpy(), logits=logits.numpy(),
             hidden=hs.numpy(), cfg=np.array([c["F"], c["H"], c["D"], 41, c["L"], c["ps"], c["st"]]),
             **extra, **arrs)


def make_autocast_forward():
    """The reference's forward under its own precision regime (rnn_args.yaml:19 use_amp: true; rnn_trainer.py:527, :704:
    torch.autocast(device_type=..., dtype=torch.bfloat16) around the model call), captured on the CPU backend: the day-layer einsum
    and the output Linear run in bf16 there, torch's CPU GRU stays fp32 (on CUDA cuDNN's GRU runs in bf16 as well).  The fp32
    logits of the same weights are stored next to them: the repository's bf16 mode has to sit as close to the autocast logits as
    the autocast logits sit to the fp32 ones (tests/test_gpu_step_parity.py)."""
    cases = [dict(tag="patch", F=48, H=96, D=5, L=5, B=3, T=61, ps=14, st=4),
             dict(tag="h256", F=64, H=256, D=3, L=2, B=4, T=40, ps=0, st=0)]
    out = {}
    for c in cases:
        torch.manual_seed(10)
        m = GRUDecoder(c["F"], c["H"], c["D"], 41, 0.0, 0.0, c["L"], c["ps"], c["st"]).eval()
        gen = torch.Generator().manual_seed(321)
        perturb_days(m, gen)
        with torch.no_grad():
            m.out.weight.mul_(4.0)     # logits of order one: margins between phonemes as in a trained model
        x = torch.randn(c["B"], c["T"], c["F"], generator=gen)
        day = torch.randint(0, c["D"], (c["B"],), generator=gen)
        with torch.no_grad():
            ref, _ = m(x, day, None, True)
            with torch.autocast(device_type="cpu", dtype=torch.bfloat16):
                amp, _ = m(x, day, None, True)
        t = c["tag"]
        out.update({f"{t}::x": x.numpy(), f"{t}::day_idx": day.numpy(), f"{t}::logits_fp32": ref.numpy(),
                    f"{t}::logits_autocast": amp.float().numpy(),
                    f"{t}::cfg": np.array([c["F"], c["H"], c["D"], 41, c["L"], c["ps"], c["st"]])})
        out.update({f"{t}::sd::{k}": v for k, v in sd_np(m).items()})
        print(t, "autocast vs fp32: max |d| %.4f of max |logit| %.3f" % (float((amp.float() - ref).abs().max()), float(ref.abs().max())))
    save("fwd_autocast.npz", **out)


def make_smooth():
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(3, 40, 16, generator=gen)
    ys = gauss_smooth(x, "cpu", 2, 100, "same").numpy()
    yv = gauss_smooth(x, "cpu", 2, 100, "valid").numpy()
    # kernel taps exactly as the reference derives them (data_augmentations.py:19-24)
    from scipy.ndimage import gaussian_filter1d
    inp = np.zeros(100, dtype=np.float32); inp[50] = 1
    gk = gaussian_filter1d(inp, 2)
    gk = gk[np.argwhere(gk > 0.01)]
    gk = np.squeeze(gk / np.sum(gk))
    # another std for generality
    x2 = torch.randn(2, 30, 8, generator=gen)
    y2 = gauss_smooth(x2, "cpu", 1, 50, "same").numpy()
    save("smooth.npz", x=x.numpy(), same=ys, valid=yv, taps=gk.astype(np.float32), x2=x2.numpy(), same_std1=y2)


def make_ctc():
    gen = torch.Generator().manual_seed(7)
    B, T, C, S = 6, 30, 41, 9
    logits = torch.randn(B, T, C, generator=gen) * 2.0
    logits.requires_grad_(True)
    targets = torch.randint(1, C, (B, S), generator=gen)
    targets[1, :4] = torch.tensor([5, 5, 5, 7])      # repeats
    targets[2, :3] = torch.tensor([3, 3, 3])
    tgt_len = torch.tensor([9, 4, 3, 1, 7, 9])
    in_len = torch.tensor([30, 30, 17, 5, 22, 19], dtype=torch.int32)
    for b in range(B):
        targets[b, tgt_len[b]:] = 0
    crit = torch.nn.CTCLoss(blank=0, reduction="none", zero_infinity=False)
    loss = crit(torch.permute(logits.log_softmax(2), [1, 0, 2]), targets, in_len, tgt_len)
    loss.mean().backward()
    save("ctc.npz", logits=logits.detach().numpy(), targets=targets.numpy(), in_len=in_len.numpy(),
         tgt_len=tgt_len.numpy(), loss=loss.detach().numpy(), dlogits=logits.grad.numpy())
    # infeasible case: T too short -> inf
    lg = torch.randn(1, 3, C, generator=gen)
    l_inf = crit(torch.permute(lg.log_softmax(2), [1, 0, 2]), torch.tensor([[4, 4, 4]]),
                 torch.tensor([3], dtype=torch.int32), torch.tensor([3]))
    save("ctc_inf.npz", logits=lg.numpy(), loss=l_inf.numpy())


def _trainer_shell(model, args):
    tr = ref_trainer.BrainToTextDecoder_Trainer.__new__(ref_trainer.BrainToTextDecoder_Trainer)
    tr.args = args
    tr.model = model
    tr.device = torch.device("cpu")
    tr.transform_args = args["dataset"]["data_transforms"]
    return tr


BASE_ARGS = dict(
    lr_max=0.005, lr_min=0.0001, lr_decay_steps=120000, lr_warmup_steps=1000,
    lr_max_day=0.005, lr_min_day=0.0001, lr_decay_steps_day=120000, lr_warmup_steps_day=1000,
    beta0=0.9, beta1=0.999, epsilon=0.1, weight_decay=0.001, weight_decay_day=0,
    grad_norm_clip_value=10, lr_scheduler_type="cosine",
    dataset=dict(data_transforms=dict(white_noise_std=1.0, constant_offset_std=0.2, random_walk_std=0.0,
                                      random_walk_axis=-1, static_gain_std=0.0, random_cut=3,
                                      smooth_kernel_size=100, smooth_data=True, smooth_kernel_std=2)),
)


def make_train_step(name="train_step.npz", ps=0, st=0, dims=(32, 48, 6, 3, 8, 40, 6), ckpt=False):
    """Four optimizer steps of the reference step body (rnn_trainer.py:513-558), dropout 0,
    noise off (val-mode transform), fused=False AdamW on CPU (same math as fused).
    ckpt: after step 3 the reference's save_model_checkpoint (rnn_trainer.py:387-406) writes ckpt_ref.pt; the state
    after the 4th step goes to ckpt_ref_step4.npz (a trainer resumed from the checkpoint must reproduce it)."""
    F_, H, D, L, B, T, S = dims
    torch.manual_seed(10)
    model = GRUDecoder(F_, H, D, 41, 0.0, 0.0, L, ps, st)
    gen = torch.Generator().manual_seed(99)
    perturb_days(model, gen)
    args = dict(BASE_ARGS)
    args["lr_warmup_steps"] = 4; args["lr_warmup_steps_day"] = 4     # non-zero lr quickly
    args["grad_norm_clip_value"] = 0.5                               # make clipping bite
    tr = _trainer_shell(model, args)
    # create_optimizer passes fused=True which needs a GPU tensor: patch AdamW to drop it
    real_adamw = torch.optim.AdamW
    torch.optim.AdamW = lambda groups, **kw: real_adamw(groups, **{k: v for k, v in kw.items() if k != "fused"})
    try:
        opt = tr.create_optimizer()
    finally:
        torch.optim.AdamW = real_adamw
    sched = tr.create_cosine_lr_scheduler(opt)
    crit = torch.nn.CTCLoss(blank=0, reduction="none", zero_infinity=False)

    x = torch.randn(B, T, F_, generator=gen)
    day = torch.tensor([0, 0, 2, 2, 2, 5, 5, 0]) if (B, D) == (8, 6) else (torch.arange(B) // 2 * 3) % D
    targets = torch.randint(1, 41, (B, S), generator=gen)
    tgt_len = torch.randint(2, S + 1, (B,), generator=gen)
    for b in range(B):
        targets[b, tgt_len[b]:] = 0
    n_steps_t = torch.randint(T - 15, T + 1, (B,), generator=gen)
    out = {f"sd0::{k}": v for k, v in sd_np(model).items()}
    out.update(x=x.numpy(), day_idx=day.numpy(), targets=targets.numpy(), tgt_len=tgt_len.numpy(),
               n_time_steps=n_steps_t.numpy(), cfg=np.array([F_, H, D, 41, L, ps, st]),
               clip=np.float32(args["grad_norm_clip_value"]), warmup=np.int64(4))
    for step in range(4):
        opt.zero_grad()
        feats, nts = tr.transform_data(x.clone(), n_steps_t, "val")
        if ps > 0:                         # rnn_trainer.py:532
            adjusted = ((nts - ps) / st + 1).to(torch.int32)
        else:
            adjusted = nts.to(torch.int32)     # patch_size 0 => adjusted_lens = n_time_steps
        logits = model(feats, day)
        loss = crit(torch.permute(logits.log_softmax(2), [1, 0, 2]), targets, adjusted, tgt_len)
        loss = torch.mean(loss)
        loss.backward()
        if step == 0:
            for n, p in model.named_parameters():
                if p.grad is not None:
                    out[f"grad0::{n}"] = p.grad.detach().numpy().copy()
            out["logits0"] = logits.detach().numpy().copy()
            out["feats0"] = feats.detach().numpy().copy()
        gn = torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=args["grad_norm_clip_value"],
                                            error_if_nonfinite=True, foreach=True)
        out[f"loss{step}"] = np.float32(loss.item())
        out[f"gnorm{step}"] = np.float32(gn.item())
        out[f"lr{step}"] = np.array([g["lr"] for g in opt.param_groups], dtype=np.float64)
        opt.step()
        sched.step()
        for k, v in sd_np(model).items():
            out[f"sd{step+1}::{k}"] = v
        if ckpt and step == 2:
            import logging
            tr.optimizer, tr.learning_rate_scheduler = opt, sched
            tr.logger = logging.getLogger("golden")
            tr.args = dict(args); tr.args["checkpoint_dir"] = "/tmp"
            tr.save_model_checkpoint(os.path.join(HERE, "ckpt_ref.pt"), 0.25, 1.5)
            print(f"wrote ckpt_ref.pt: {os.path.getsize(os.path.join(HERE, 'ckpt_ref.pt'))/1e6:.2f} MB")
    if ckpt:
        save("ckpt_ref_step4.npz", **{k: v for k, v in out.items() if k.startswith("sd4::") or k in
                                      ("x", "day_idx", "targets", "tgt_len", "n_time_steps", "cfg", "clip", "warmup", "loss3", "gnorm3", "lr3")})
        return
    save(name, **out)


def make_lr_table():
    model = GRUDecoder(8, 8, 2, 41, 0, 0, 1, 0, 0)
    tr = _trainer_shell(model, dict(BASE_ARGS))
    real_adamw = torch.optim.AdamW
    torch.optim.AdamW = lambda groups, **kw: real_adamw(groups, **{k: v for k, v in kw.items() if k != "fused"})
    try:
        opt = tr.create_optimizer()
    finally:
        torch.optim.AdamW = real_adamw
    sched = tr.create_cosine_lr_scheduler(opt)
    steps = [0, 1, 2, 999, 1000, 1001, 60000, 119999, 120000, 130000]
    fac = np.array([[lam(s) for lam in sched.lr_lambdas] for s in steps], dtype=np.float64)
    groups = np.array([g["group_type"] for g in opt.param_groups])
    # observed lr of the first optimizer steps (LambdaLR(...,-1): batch i uses f(i))
    lrs = []
    for i in range(3):
        lrs.append([g["lr"] for g in opt.param_groups])
        opt.step(); sched.step()
    # lr_scheduler_type 'linear' (rnn_trainer.py:228-234): torch LinearLR over the same three groups
    targs = dict(BASE_ARGS); targs["lr_decay_steps"] = 50
    tr2 = _trainer_shell(GRUDecoder(8, 8, 2, 41, 0, 0, 1, 0, 0), targs)
    torch.optim.AdamW = lambda groups, **kw: real_adamw(groups, **{k: v for k, v in kw.items() if k != "fused"})
    try:
        opt2 = tr2.create_optimizer()
    finally:
        torch.optim.AdamW = real_adamw
    lin = torch.optim.lr_scheduler.LinearLR(optimizer=opt2, start_factor=1.0, end_factor=targs["lr_min"] / targs["lr_max"],
                                            total_iters=targs["lr_decay_steps"])
    lin_lrs = []
    for i in range(60):
        lin_lrs.append([g["lr"] for g in opt2.param_groups])
        opt2.step(); lin.step()
    save("lr_table.npz", steps=np.array(steps), factors=fac, groups=groups, first_lrs=np.array(lrs),
         linear_lrs=np.array(lin_lrs, dtype=np.float64), linear_total=np.int64(50),
         linear_sd_keys=np.array(sorted(lin.state_dict().keys())), lambda_sd_keys=np.array(sorted(sched.state_dict().keys())),
         optim_group_keys=np.array(sorted(opt.state_dict()["param_groups"][0].keys())))


def make_transform():
    model = GRUDecoder(8, 8, 2, 41, 0, 0, 1, 0, 0)
    tr = _trainer_shell(model, dict(BASE_ARGS))
    gen = torch.Generator().manual_seed(11)
    B, T, C = 3, 25, 12
    x = torch.randn(B, T, C, generator=gen)
    # zero-padded tail on trial 1 (n_time_steps shorter)
    n = torch.tensor([25, 18, 25])
    x[1, 18:] = 0
    wn = torch.randn(B, T, C, generator=gen)
    on = torch.randn(B, 1, C, generator=gen)
    out = dict(x=x.numpy(), n_time_steps=n.numpy(), white=wn.numpy(), offset=on.numpy())
    real_randn = torch.randn
    real_randint = np.random.randint
    for cut in (0, 1, 2):
        draws = [wn, on]

        def fake_randn(*shape, **kw):
            return draws.pop(0).clone()
        torch.randn = fake_randn
        np.random.randint = lambda lo, hi=None, **kw: cut
        try:
            y, nn_ = tr.transform_data(x.clone(), n.clone(), "train")
        finally:
            torch.randn = real_randn
            np.random.randint = real_randint
        out[f"train_cut{cut}"] = y.numpy()
        out[f"train_cut{cut}_n"] = nn_.numpy()
    yv, nv = tr.transform_data(x.clone(), n.clone(), "val")
    out["val"] = yv.numpy(); out["val_n"] = nv.numpy()
    # every augmentation on (static gain rnn_trainer.py:449-453, random walk :464-465), draws injected in call order
    for axis in (-1, 1):
        a2 = dict(BASE_ARGS)
        a2["dataset"] = dict(data_transforms=dict(BASE_ARGS["dataset"]["data_transforms"], static_gain_std=0.1,
                                                  random_walk_std=0.05, random_walk_axis=axis))
        tr2 = _trainer_shell(model, a2)
        sg = torch.randn(B, C, C, generator=gen)
        rw = torch.randn(B, T, C, generator=gen)
        draws = [wn, on, rw]
        real_like = torch.randn_like
        torch.randn = lambda *shape, **kw: draws.pop(0).clone()
        torch.randn_like = lambda t, **kw: sg.clone()
        np.random.randint = lambda lo, hi=None, **kw: 1
        try:
            y, nn_ = tr2.transform_data(x.clone(), n.clone(), "train")
        finally:
            torch.randn = real_randn; torch.randn_like = real_like; np.random.randint = real_randint
        tag = "last" if axis == -1 else "time"
        out[f"full_{tag}"] = y.numpy(); out[f"full_{tag}_sg"] = sg.numpy(); out[f"full_{tag}_rw"] = rw.numpy()
    save("transform.npz", **out)


def make_greedy():
    gen = torch.Generator().manual_seed(21)
    B, T, C = 5, 33, 41
    # peaky logits so that blanks/repeats occur
    base = torch.randn(B, T, C, generator=gen)
    base[..., 0] += 1.5
    base = torch.repeat_interleave(base[:, ::3], 3, dim=1)[:, :T] + 0.05 * torch.randn(B, T, C, generator=gen)
    lens = torch.tensor([33, 20, 33, 7, 29])
    labels = torch.randint(1, C, (B, 8), generator=gen)
    lab_len = torch.tensor([8, 5, 8, 2, 6])
    out = dict(logits=base.numpy(), lens=lens.numpy(), labels=labels.numpy(), lab_len=lab_len.numpy())
    import torch.nn.functional as Fnn  # noqa
    for b in range(B):
        # trainer rule (rnn_trainer.py:725-728)
        d = torch.argmax(base[b, 0:lens[b], :], dim=-1)
        d = torch.unique_consecutive(d, dim=-1).numpy()
        d = np.array([i for i in d if i != 0], dtype=np.int64)
        out[f"trainer_{b}"] = d
        out[f"edit_{b}"] = np.int64(_edit_distance(d, labels[b][:lab_len[b]].numpy()))
        # evaluate rule (evaluate_model.py:131-137)
        p = np.argmax(base[b].numpy(), axis=-1)
        p = [int(q) for q in p if q != 0]
        p = [p[i] for i in range(len(p)) if i == 0 or p[i] != p[i - 1]]
        out[f"evaluate_{b}"] = np.array(p, dtype=np.int64)
    out["rearranged"] = ref_helpers.rearrange_speech_logits_pt(base.numpy())
    save("greedy.npz", **out)


def make_evalstep():
    """runSingleDecodingStep with fp32 input (evaluate_model_helpers.py:87-115)."""
    F_, H, D, L = 40, 64, 3, 5
    torch.manual_seed(10)
    m = GRUDecoder(F_, H, D, 41, 0.4, 0.2, L, 14, 4).eval()
    gen = torch.Generator().manual_seed(31)
    perturb_days(m, gen)
    x = torch.randn(1, 90, F_, generator=gen)
    margs = dict(use_amp=False, dataset=dict(data_transforms=dict(smooth_kernel_std=2, smooth_kernel_size=100)))
    logits = ref_helpers.runSingleDecodingStep(x, 1, m, margs, "cpu")
    arrs = {f"sd::{k}": v for k, v in sd_np(m).items()}
    save("evalstep.npz", x=x.numpy(), day=np.int64(1), logits=logits,
         cfg=np.array([F_, H, D, 41, L, 14, 4]), **arrs)
    cases = ["Hello, World!", "it's  a  -- test - case", "A 'quoted' word.", "  spaces   everywhere  "]
    with open(os.path.join(HERE, "remove_punctuation.txt"), "w") as f:
        for c in cases:
            f.write(c + "\t" + ref_helpers.remove_punctuation(c) + "\n")


def make_init():
    """Initial weights of the reference class under a fixed seed (rnn_model.py:50-86 init order)."""
    for tag, cfg in (("a", (16, 32, 3, 41, 0.4, 0.2, 2, 14, 4)), ("b", (24, 48, 2, 41, 0.0, 0.0, 3, 0, 0))):
        torch.manual_seed(10)
        m = GRUDecoder(*cfg)
        arrs = {f"sd::{k}": v for k, v in sd_np(m).items()}
        names = np.array([n for n, _ in m.named_parameters()])
        save(f"init_{tag}.npz", cfg=np.array(cfg, dtype=np.float64), names=names, **arrs)


def make_sampler_index():
    """Batch index of the reference sampler under a seed (dataset.py:162-242): day-balanced random training batches and
    the sequential test batches.  h5py is stubbed: the index is built in __init__ without touching files."""
    import dataset as ref_ds
    rng = np.random.RandomState(5)
    trial_idx = {d: {"trials": sorted(rng.choice(200, size=int(rng.randint(5, 40)), replace=False).tolist()),
                     "session_path": f"/nonexistent/day{d}.hdf5"} for d in range(7)}
    out = dict(days=np.array(list(trial_idx.keys())))
    for d, v in trial_idx.items():
        out[f"trials_{d}"] = np.array(v["trials"])
    for tag, kw in (("a", dict(batch_size=10, days_per_batch=3, must_include_days=None)),
                    ("b", dict(batch_size=16, days_per_batch=4, must_include_days=[1, -1]))):
        tr = ref_ds.BrainToTextDataset(trial_idx, n_batches=12, split="train", random_seed=7, **kw)
        for bi, batch in tr.batch_index.items():
            out[f"train_{tag}_{bi}_days"] = np.array([int(d) for d in batch.keys()])
            for d, t in batch.items():
                out[f"train_{tag}_{bi}_{int(d)}"] = np.asarray(t)
    te = ref_ds.BrainToTextDataset(trial_idx, n_batches=None, split="test", batch_size=16, random_seed=7)
    out["test_n"] = np.int64(len(te.batch_index))
    for bi, batch in te.batch_index.items():
        (d, t), = batch.items()
        out[f"test_{bi}_day"] = np.int64(d); out[f"test_{bi}"] = np.asarray(t)
    save("sampler_index.npz", **out)


def make_dataset_batches():
    """Batch dicts of the reference's BrainToTextDataset.__getitem__ (dataset.py:100-159: per-day file reads, feature
    subset, pad_sequence, dtypes, key set, a listed-but-missing trial skipped) on an in-memory h5py stand-in: two random
    train batches, the sequential test batches, and a train batch with feature_subset.  The fixture carries the trial
    store itself so that the tests can serve the same "files" to the product's dataset and to ResidentDataset."""
    import dataset as ref_ds
    rng = np.random.RandomState(11)
    F, LAB, TR = 32, 20, 24
    store = {}
    trial_idx = {}
    for d in range(3):
        path = f"/mem/day{d}/data_train.hdf5"
        ids = sorted(rng.choice(60, size=int(rng.randint(9, 14)), replace=False).tolist())
        trial_idx[d] = {"trials": ids, "session_path": path}
        MEMFILES[path] = {}
        for t in ids:
            if d == 1 and t == ids[2]:
                continue                      # listed in the index, absent from the file: printed and skipped (dataset.py:144-146)
            T = int(rng.randint(17, 49)); S = int(rng.randint(3, 12))
            lab = np.zeros(LAB, np.int32); lab[:S] = rng.randint(1, 41, size=S)
            g = dict(input_features=rng.randn(T, F).astype(np.float32), seq_class_ids=lab,
                     transcription=rng.randint(0, 128, size=TR).astype(np.int32),
                     attrs=dict(n_time_steps=np.int64(T), seq_len=np.int64(S), block_num=np.int64(rng.randint(1, 9)),
                                trial_num=np.int64(t)))
            MEMFILES[path][f"trial_{t:04d}"] = g
            for k in ("input_features", "seq_class_ids", "transcription"):
                store[f"store_{d}_{t}_{k}"] = g[k]
            store[f"store_{d}_{t}_attrs"] = np.array([g["attrs"][k] for k in ("n_time_steps", "seq_len", "block_num", "trial_num")], np.int64)
    out = dict(store)
    out["days"] = np.arange(3)
    for d, v in trial_idx.items():
        out[f"trials_{d}"] = np.array(v["trials"])
    KEYS = ("input_features", "seq_class_ids", "n_time_steps", "phone_seq_lens", "day_indicies", "transcriptions", "block_nums", "trial_nums")

    def dump(tag, ds, n):
        out[f"{tag}_n"] = np.int64(n)
        for bi in range(n):
            b = ds[bi]
            assert tuple(b.keys()) == KEYS
            for k in KEYS:
                out[f"{tag}_{bi}_{k}"] = b[k].numpy()
                out[f"{tag}_{bi}_{k}_dtype"] = np.array(str(b[k].dtype))
            idx = ds.batch_index[bi]
            out[f"{tag}_{bi}_index_days"] = np.array([int(d) for d in idx.keys()])
            for d, t in idx.items():
                out[f"{tag}_{bi}_index_{int(d)}"] = np.asarray(t)

    tr = ref_ds.BrainToTextDataset(trial_idx, n_batches=2, split="train", batch_size=8, days_per_batch=2, random_seed=3)
    dump("train", tr, 2)
    te = ref_ds.BrainToTextDataset(trial_idx, n_batches=None, split="test", batch_size=8, random_seed=3)
    dump("test", te, len(te.batch_index))
    sub = [3, 1, 30, 7, 8, 9, 10, 31]
    trs = ref_ds.BrainToTextDataset(trial_idx, n_batches=1, split="train", batch_size=6, days_per_batch=3, random_seed=4,
                                    feature_subset=sub)
    out["subset"] = np.array(sub)
    dump("trainsub", trs, 1)
    save("dataset_batches.npz", **out)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "dataset":
        make_dataset_batches()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "init":
        make_init()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "autocast":
        make_autocast_forward()
        sys.exit(0)
    make_init()
    make_forward()
    make_autocast_forward()
    make_smooth()
    make_ctc()
    make_train_step()
    make_train_step("train_step_patch.npz", ps=14, st=4, dims=(16, 48, 5, 3, 6, 73, 5))
    make_train_step(ckpt=True, ps=14, st=4, dims=(16, 32, 4, 2, 6, 61, 4))
    make_sampler_index()
    make_dataset_batches()
    make_lr_table()
    make_transform()
    make_greedy()
    make_evalstep()
