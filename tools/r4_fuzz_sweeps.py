#!/usr/bin/env python3
"""Random shapes through every sweep variant: persistent fp32 (device scope / XCD-local, 16- and 32-unit workgroups), the fused
forward, bf16 operands (16-unit; 32-unit with fp32 tiles and with the fragment hand-off) against the step-launch kernels (fp32:
3e-6 of the max; bf16 variants among themselves: bit-identical where the kernels promise it).  usage: r4_fuzz_sweeps.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nejm-brain-to-text_amd"))
import numpy as np, torch
import b2t_native as Nn, b2t_ops as ops
lib = Nn.load(); dev = torch.device("cuda:0"); p = ops._p
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
LOCAL, PARITY, WIDE, BF16 = 0x400, 0x800, 0x200, 0x100
bad = 0
for case in range(N):
    H = int(rng.choice([16, 32, 48, 64, 80, 96, 128, 160, 192, 256, 272, 288, 320, 384, 512, 544, 640, 768]))
    B = int(rng.randint(1, 71)); T = int(rng.randint(1, 34)) if rng.rand() > 0.12 else int(rng.randint(90, 131))   # (long calls: the fragment buffer wraps around)
    if (H // 16) * ((B + 15) // 16) > 256: continue
    g = torch.Generator().manual_seed(case * 7 + 1)
    rnd = lambda *s: (torch.randn(*s, generator=g)).to(dev)
    gi, w, b_, h0 = rnd(T, B, 3 * H) * 0.5, rnd(3 * H, H) * (1.0 / H ** 0.5), rnd(3 * H) * 0.1, rnd(B, H) * 0.3
    dY, dhl = rnd(T, B, H) * 0.05, rnd(B, H) * 0.05
    w2, b2 = rnd(3 * H, H) * (1.0 / H ** 0.5), rnd(3 * H) * 0.1
    wt = w.t().contiguous()
    sync = torch.zeros(lib.b2t_gru_ws_bytes(T, B, H) // 4 + 16, dtype=torch.int32, device=dev)
    def run(mode, env=None, fused=False):
        old = {k: os.environ.get(k) for k in (env or {})}
        os.environ.update(env or {})
        try:
            out = torch.zeros(T + 1, B, H, device=dev); out[0] = h0
            res = torch.zeros(T, B, 4 * H, device=dev)
            gi2 = torch.zeros(T, B, 3 * H, device=dev)
            if fused:
                Nn.check(lib.b2t_gru_layer_fwd_fused_f32(p(gi), p(w), p(b_), p(out[0]), p(out[1:]), p(res), None, p(w2), p(b2), p(gi2), T, B, H, mode, p(sync), ops._stream()), "fused")
            else:
                Nn.check(lib.b2t_gru_layer_fwd_f32(p(gi), p(w), p(b_), p(out[0]), p(out[1:]), p(res), None, T, B, H, mode, p(sync), ops._stream()), "fwd")
            dG = torch.zeros(T, B, 4 * H, device=dev); dh = torch.zeros(B, H, device=dev); sc = torch.empty(B, H, device=dev)
            Nn.check(lib.b2t_gru_layer_bwd_f32(p(dY), p(dhl), p(res), p(out[1:]), p(out[0]), p(wt), p(dG), p(dh), p(sc), T, B, H, mode, p(sync), ops._stream()), "bwd")
            torch.cuda.synchronize()
            assert int(sync[0]) == 0, "hand-off timeout"
            return out, res, dG, dh, gi2
        finally:
            for k, v in old.items():
                if v is None: os.environ.pop(k, None)
                else: os.environ[k] = v
    def close(a, r, tol):
        return all(float((x - y).abs().max()) <= tol * max(1.0, float(y.abs().max())) for x, y in zip(a[:4], r[:4]))
    def same(a, r, upto=4):
        return all(torch.equal(x, y) for x, y in zip(a[:upto], r[:upto]))
    ref = run(0)
    base = run(1)
    msgs = []
    if not close(base, ref, 3e-6): msgs.append("persistent != step-launch")
    for extra in (LOCAL, LOCAL | PARITY):
        if not same(run(1 | extra), base): msgs.append(f"local {extra:#x} != device scope")
    if H % 32 == 0 and H <= 512:
        wd = run(1 | WIDE)
        if not close(wd, ref, 3e-6): msgs.append("wide fp32 != step-launch")
        if not same(run(1 | WIDE | LOCAL), wd): msgs.append("wide local != wide")
    if H <= 512:
        fu = run(1, fused=True)
        if not same(fu, base): msgs.append("fused recurrence != plain")
        proj = torch.einsum("tbh,gh->tbg", base[0][1:], w2) + b2
        if float((fu[4] - proj).abs().max()) > 3e-5 * max(1.0, float(proj.abs().max())): msgs.append("fused projection off")
        if not same(run(1 | LOCAL, fused=True), fu, 4) : msgs.append("fused local != fused")
    bn = run(1 | BF16)
    if not close(bn, ref, 3e-2): msgs.append("bf16 far from fp32")
    if H % 32 == 0 and H <= 768:
        t0 = run(1 | BF16 | WIDE, {"B2T_HANDOFF16": "0"})
        if not same(t0[:2], bn[:2], 2): msgs.append("bf16 wide fwd != bf16 narrow fwd")
        for extra in (0, LOCAL, LOCAL | PARITY):
            if not same(run(1 | BF16 | WIDE | extra), t0): msgs.append(f"bf16 fragments {extra:#x} != fp32 tiles")
    if msgs:
        bad += 1
        print(f"case {case}: B={B} H={H} T={T}: " + "; ".join(msgs), flush=True)
print(f"{N} cases, {bad} with differences")
