#!/usr/bin/env python3
"""XCD-local vs device-scope hand-off of the persistent sweeps on small / odd shapes: max |difference| per output (must be 0)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nejm-brain-to-text_amd"))
import torch
import b2t_native as Nn, b2t_ops as ops
lib = Nn.load(); dev = torch.device("cuda:0"); p = ops._p
for (B, H, T) in [(5, 48, 9), (17, 80, 13), (33, 272, 11), (64, 512, 24), (64, 512, 125), (16, 16, 5), (64, 128, 20)]:
    g = torch.Generator().manual_seed(B * 1000 + H)
    rnd = lambda *s: (torch.randn(*s, generator=g)).to(dev)
    gi, w, b_, h0 = rnd(T, B, 3 * H) * 0.5, rnd(3 * H, H) * (1.0 / H ** 0.5), rnd(3 * H) * 0.1, rnd(B, H) * 0.3
    dY, dhl = rnd(T, B, H) * 0.05, rnd(B, H) * 0.05
    wt = w.t().contiguous()
    def run(mode):
        out = torch.zeros(T + 1, B, H, device=dev); out[0] = h0
        res = torch.zeros(T, B, 4 * H, device=dev)
        sync = torch.zeros(lib.b2t_gru_ws_bytes(T, B, H) // 4 + 16, dtype=torch.int32, device=dev)
        Nn.check(lib.b2t_gru_layer_fwd_f32(p(gi), p(w), p(b_), p(out[0]), p(out[1:]), p(res), None, T, B, H, mode, p(sync), ops._stream()), "fwd")
        dG = torch.zeros(T, B, 4 * H, device=dev); dh = torch.zeros(B, H, device=dev); sc = torch.empty(B, H, device=dev)
        Nn.check(lib.b2t_gru_layer_bwd_f32(p(dY), p(dhl), p(res), p(out[1:]), p(out[0]), p(wt), p(dG), p(dh), p(sc), T, B, H, mode, p(sync), ops._stream()), "bwd")
        torch.cuda.synchronize()
        return out, res, dG, dh, int(sync[0])
    base = run(1)
    ref0 = run(0)
    line = [f"B={B} H={H} T={T}: persistent vs step-launch out {float((base[0] - ref0[0]).abs().max()):.1e}"]
    for extra, nm in ((0x400, "local"), (0xC00, "local+parity")):
        for rep in range(3):
            got = run(1 | extra)
            d = [float((a - r).abs().max()) for a, r in zip(got[:4], base[:4])]
            line.append(f"{nm}#{rep} out {d[0]:.1e} res {d[1]:.1e} dG {d[2]:.1e} dh {d[3]:.1e} err {got[4]}")
    print(" | ".join(line))
