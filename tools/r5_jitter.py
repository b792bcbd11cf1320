#!/usr/bin/env python3
"""Round-5 verdict 1c: hunt for an ordering the eager plan has only by timing.  B2T_EXEC_JITTER=seed makes csrc/exec.cpp enqueue a
10-400 us spin kernel in front of a random third of a pass's tasks; forward + CTC + backward are run N times with different seeds
and every gradient / loss must equal the unjittered plan's bit for bit.  Also runs a jittered TRAINING trajectory (the step incl.
clip + AdamW) at BASELINE configs[1] against the unjittered one.
usage: r5_jitter.py [n_small=200] [n_c2=200] [n_traj=30]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "nejm-brain-to-text_amd"))
import torch
import bench, b2t_ops as ops
from rnn_model import GRUDecoder
from b2t_train_step import TrainStep

n_small = int(sys.argv[1]) if len(sys.argv) > 1 else 200
n_c2 = int(sys.argv[2]) if len(sys.argv) > 2 else 200
n_traj = int(sys.argv[3]) if len(sys.argv) > 3 else 30
dev = torch.device("cuda:0")


def hunt(tag, F, H, D, C, L, B, T, S, chunks, chunks_bwd, mask, n, patch=(0, 0), amp=False):
    os.environ.pop("B2T_EXEC_JITTER", None)
    ops.PIPELINE["chunks"], ops.PIPELINE["chunks_bwd"], ops.PIPELINE["wgrad_chunk_mask"] = chunks, chunks_bwd, mask
    old_amp = ops.AMP["on"]
    ops.set_amp(amp)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, T, F, generator=g).to(dev)
    day = torch.randint(0, D, (B,), generator=g)
    tgt = torch.randint(1, C, (B, S), generator=g); tl = torch.randint(3, S + 1, (B,), generator=g)
    nt = torch.randint(T - 40, T + 1, (B,), generator=g)
    for b in range(B):
        tgt[b, tl[b]:] = 0
    torch.manual_seed(3)
    m = GRUDecoder(F, H, D, C, 0.0, 0.0, L, patch[0], patch[1]).to(dev).train()
    ts = TrainStep(m, dict(bench.ARGS))
    loss = ts.compute_grads(x, day, tgt, nt, tl).clone()
    torch.cuda.synchronize()
    ref, logits = ts.grad_arena.clone(), ts.last_logits.clone()
    bad = []
    for seed in range(n):
        os.environ["B2T_EXEC_JITTER"] = str(seed)
        ts.grad_arena.zero_()
        l2 = ts.compute_grads(x, day, tgt, nt, tl)
        torch.cuda.synchronize()
        if not (torch.equal(l2, loss) and torch.equal(ts.grad_arena, ref) and torch.equal(ts.last_logits, logits)):
            d = (ts.grad_arena - ref).abs()
            bad.append((seed, float(d.max()), int((d > 0).sum()), float((l2 - loss).abs().max())))
    ts.check_status()
    ops.set_amp(old_amp)
    os.environ.pop("B2T_EXEC_JITTER", None)
    print(f"{tag}: {n} jitter seeds, {len(bad)} differ from the unjittered plan {bad[:5]}", flush=True)
    return bad


def trajectory(n):
    ops.PIPELINE["chunks"], ops.PIPELINE["chunks_bwd"], ops.PIPELINE["wgrad_chunk_mask"] = DEF
    outs = []
    for jitter in (False, True):
        os.environ.pop("B2T_EXEC_JITTER", None)
        torch.manual_seed(10)
        model = GRUDecoder(bench.F, bench.H, bench.D, bench.C, 0.0, 0.0, bench.L, 0, 0).to(dev).train()
        ts = TrainStep(model, dict(bench.ARGS))
        x, days, labels, nts, lens = bench.make_batch(1000, dev)
        losses = []
        for i in range(n):
            if jitter:
                os.environ["B2T_EXEC_JITTER"] = str(1000 + i)
            f = ops.augment_smooth(x, 2, 100, "same", cut=i % 3, white_std=1.0, offset_std=0.2, seed=i)
            losses.append(ts.step(f, days, labels, nts - i % 3, lens)[0])
        torch.cuda.synchronize()
        ts.check_status()
        outs.append(([float(l) for l in losses], model.arena().clone()))
    os.environ.pop("B2T_EXEC_JITTER", None)
    same = outs[0][0] == outs[1][0] and (outs[0][1] is None or torch.equal(outs[0][1], outs[1][1]))
    print(f"C2 training trajectory, {n} steps, every pass jittered: losses {'IDENTICAL' if same else 'DIFFER'} "
          f"(last {outs[0][0][-1]:.6f} / {outs[1][0][-1]:.6f})", flush=True)
    return same


DEF = (ops.PIPELINE["chunks"], ops.PIPELINE["chunks_bwd"], ops.PIPELINE["wgrad_chunk_mask"])
bad = []
if len(sys.argv) > 4 and sys.argv[4] == "amp":
    # the bf16 plans of round 5 (pre-packed weights, xpack tasks incl. the day layer's, Z-batched day GEMMs, 256-tile kernel where it fits)
    bad += hunt("bf16, patch 4/2 (H=256, L=3, F=512, B=32, T=208, chunks 3/2)", 512, 256, 6, 41, 3, 32, 208, 12, 3, 2, 0, n_small, patch=(4, 2), amp=True)
    bad += hunt("bf16, no patch (H=256, L=3, F=512, B=32, T=208, chunks 3/2)", 512, 256, 6, 41, 3, 32, 208, 12, 3, 2, 0, n_small, amp=True)
    bad += hunt("bf16, C2 default plan", bench.F, bench.H, bench.D, bench.C, bench.L, bench.B, bench.T, bench.S, *DEF, n_c2, amp=True)
    print("RESULT:", "clean" if not bad else "HAZARD FOUND")
    sys.exit(0 if not bad else 1)
for mask in (0, 0b111):
    bad += hunt(f"small (H=128, L=3, B=32, T=160, chunks 5/3, wgrad mask {mask})", 64, 128, 6, 41, 3, 32, 160, 12, 5, 3, mask, n_small // 2)
bad += hunt(f"C2 (H=512, L=5, B=64, T=500, default plan {DEF})", bench.F, bench.H, bench.D, bench.C, bench.L, bench.B, bench.T, bench.S, *DEF, n_c2)
ok = trajectory(n_traj)
print("RESULT:", "clean" if not bad and ok else "HAZARD FOUND")
sys.exit(0 if not bad and ok else 1)
