"""The reference's training step on PyTorch-CPU operators — TEST INFRASTRUCTURE and bench.py's `cpu_baseline` leg only.

Nothing under nejm-brain-to-text_amd/ imports this.  It is the operator sequence of the reference's step body
(model_training/rnn_trainer.py:527-558 with rnn_model.py:88-134 and data_augmentations.py:6-37) written against plain
torch ops — `F.conv1d` smoothing, `einsum` day layer + Softsign, `unfold` patching, `nn.GRU`, `nn.Linear`, `log_softmax` +
`nn.CTCLoss('none')` + mean, `clip_grad_norm_`, `AdamW` (3 groups), cosine `LambdaLR` — because the reference's own files
do not travel to the GPU box.  tests/test_oracle_golden.py pins it to tests/golden/train_step*.npz (captured
by importing the reference), so it is a verified restatement, not a second opinion.
"""
import math

import numpy as np
import torch
import torch.nn.functional as Fnn


def gauss_taps(std=2.0, size=100):
    """data_augmentations.py:19-24 without scipy: gaussian_filter1d's truncated (4 sigma) kernel on a unit impulse."""
    r = int(4.0 * std + 0.5)
    xs = np.arange(-r, r + 1, dtype=np.float64)
    phi = np.exp(-0.5 / (std * std) * xs * xs); phi /= phi.sum()
    resp = np.zeros(size, dtype=np.float32); c = size // 2
    resp[c - r:c + r + 1] = phi.astype(np.float32)
    keep = resp[resp > 0.01]
    return torch.from_numpy((keep / keep.sum()).astype(np.float32))


class CpuGRUDecoder(torch.nn.Module):
    """rnn_model.py:4-134 (same parameter names, so a reference state_dict loads)."""

    def __init__(self, F, H, D, C, L, patch_size=0, patch_stride=0):
        super().__init__()
        self.F, self.H, self.L, self.ps, self.st = F, H, L, patch_size, patch_stride
        self.day_weights = torch.nn.ParameterList([torch.nn.Parameter(torch.eye(F)) for _ in range(D)])
        self.day_biases = torch.nn.ParameterList([torch.nn.Parameter(torch.zeros(1, F)) for _ in range(D)])
        self.gru = torch.nn.GRU(F * patch_size if patch_size > 0 else F, H, num_layers=L, batch_first=True)
        self.out = torch.nn.Linear(H, C)
        self.h0 = torch.nn.Parameter(torch.zeros(1, 1, H))

    def forward(self, x, day_idx):
        W = torch.stack([self.day_weights[int(i)] for i in day_idx], 0)
        b = torch.cat([self.day_biases[int(i)] for i in day_idx], 0).unsqueeze(1)
        x = torch.nn.functional.softsign(torch.einsum("btd,bdk->btk", x, W) + b)
        if self.ps > 0:
            x = x.unsqueeze(1).permute(0, 3, 1, 2)
            x = x.unfold(3, self.ps, self.st).squeeze(2).permute(0, 2, 3, 1)
            x = x.reshape(x.size(0), x.size(1), -1)
        h0 = self.h0.expand(self.L, x.shape[0], self.H).contiguous()
        out, _ = self.gru(x, h0)
        return self.out(out)


def lr_factor(step, r, decay, warm):
    if step < warm:
        return float(step) / float(max(1, warm))
    if step < decay:
        prog = float(step - warm) / float(max(1, decay - warm))
        return max(r, r + (1 - r) * 0.5 * (1.0 + math.cos(math.pi * prog)))
    return r


class CpuTrainer:
    def __init__(self, model, args):
        self.model, self.a = model, args
        named = list(model.named_parameters())
        bias = [p for n, p in named if "gru.bias" in n or "out.bias" in n]
        day = [p for n, p in named if "day_" in n]
        other = [p for n, p in named if "day_" not in n and "gru.bias" not in n and "out.bias" not in n]
        a = args
        self.opt = torch.optim.AdamW([dict(params=bias, weight_decay=0), dict(params=day, lr=a["lr_max_day"], weight_decay=a["weight_decay_day"]),
                                      dict(params=other)], lr=a["lr_max"], betas=(a["beta0"], a["beta1"]), eps=a["epsilon"],
                                     weight_decay=a["weight_decay"])
        main = lambda s: lr_factor(s, a["lr_min"] / a["lr_max"], a["lr_decay_steps"], a["lr_warmup_steps"])
        dayf = lambda s: lr_factor(s, a["lr_min_day"] / a["lr_max_day"], a["lr_decay_steps_day"], a["lr_warmup_steps_day"])
        self.sched = torch.optim.lr_scheduler.LambdaLR(self.opt, [main, dayf, main], -1)
        self.ctc = torch.nn.CTCLoss(blank=0, reduction="none", zero_infinity=False)
        self.taps = gauss_taps(2.0, 100)

    def smooth(self, x):
        C = x.shape[2]
        k = self.taps.view(1, 1, -1).repeat(C, 1, 1)
        return Fnn.conv1d(x.permute(0, 2, 1), k, padding="same", groups=C).permute(0, 2, 1)

    def step(self, x, day_idx, targets, n_time_steps, tgt_len, white=None, offset=None, cut=0):
        """One step: augmentation (optional draws) -> smoothing -> forward -> CTC -> backward -> clip -> AdamW -> LR."""
        m = self.model
        self.opt.zero_grad()
        n = n_time_steps
        if white is not None:
            x = x + white * 1.0
        if offset is not None:
            x = x + offset * 0.2
        if cut > 0:
            x = x[:, cut:, :]; n = n - cut
        feats = self.smooth(x)
        adj = ((n - m.ps) / m.st + 1).to(torch.int32) if m.ps > 0 else n.to(torch.int32)
        logits = m(feats, day_idx)
        loss = self.ctc(torch.permute(logits.log_softmax(2), [1, 0, 2]), targets, adj, tgt_len).mean()
        loss.backward()
        clip = self.a["grad_norm_clip_value"]
        gn = torch.nn.utils.clip_grad_norm_(m.parameters(), max_norm=clip, error_if_nonfinite=True) if clip > 0 else torch.tensor(0.0)
        self.opt.step()
        self.sched.step()
        return float(loss.detach()), float(gn)
