#!/usr/bin/env python3
"""Fused streaming frame (csrc/stream.hip) against the executor path, and its time per frame."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "nejm-brain-to-text_amd"))
import numpy as np, torch
import b2t_native as N, b2t_ops as ops
from rnn_model import GRUDecoder
dev = torch.device("cuda:0")


def run(B, F, H, L, patch, stride, frames, per_call=1, days=4, seed=0):
    torch.manual_seed(seed)
    model = GRUDecoder(F, H, days, 41, 0.0, 0.0, L, patch, stride).to(dev).eval()
    day = (torch.arange(B, dtype=torch.int32, device=dev) % days)
    T_all = (patch if patch else 1) + (stride if patch else 1) * (frames * per_call - 1)
    x_all = torch.randn(B, T_all, F, device=dev) * 0.5
    outs = {}
    for fused in (False, True):
        ops.STREAM["fused"] = fused
        states, got = None, []
        with torch.no_grad():
            for f in range(frames):
                if patch:
                    t0 = f * per_call * stride
                    xf = x_all[:, t0: t0 + patch + stride * (per_call - 1)].contiguous()
                else:
                    xf = x_all[:, f * per_call:(f + 1) * per_call].contiguous()
                logits, states = model(xf, day, states, True)
                got.append(logits)
        torch.cuda.synchronize()
        outs[fused] = (torch.cat(got, 1), states)
    dl = float((outs[True][0] - outs[False][0]).abs().max()); dh = float((outs[True][1] - outs[False][1]).abs().max())
    print(f"B={B} F={F} H={H} L={L} patch={patch}/{stride} frames={frames}x{per_call}: max|dlogits| {dl:.2e} max|dhidden| {dh:.2e} "
          f"(|logits| max {float(outs[False][0].abs().max()):.2f}) nan={bool(torch.isnan(outs[True][0]).any())}", flush=True)
    return model, day, x_all


def timing(B=32, F=512, H=768, L=5, patch=14, stride=4, n=200):
    model, day, x_all = run(B, F, H, L, patch, stride, 3)
    xf = x_all[:, :patch].contiguous()
    for fused in (False, True):
        ops.STREAM["fused"] = fused
        states = None
        with torch.no_grad():
            for _ in range(20):
                logits, states = model(xf, day, states, True)
            torch.cuda.synchronize()
            ts = []
            for _ in range(n):
                t0 = time.perf_counter()
                logits, states = model(xf, day, states, True)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            # device time alone: n back-to-back calls between two events
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                logits, states = model(xf, day, states, True)
            e1.record(); torch.cuda.synchronize()
        ts = np.array(ts) * 1e3
        if fused:
            st = model._ws.get("stream_sync", (N.load().b2t_stream_sync_bytes() // 4,), dev, torch.int32).cpu().numpy()[32 * 18:32 * 18 + 40].astype(np.int64)
            print("phase stamps (us since kernel start):", [round(float((v - st[0]) & 0xffffffff) / 100.0, 1) for v in st[:15]])
        if fused:
            print("layer-1 product stamps:", [round(float((v - st[0]) & 0xffffffff) / 100.0, 1) for v in st[32:36]])
        print(f"fused={fused}: call+sync p50 {np.percentile(ts, 50):.4f} ms p95 {np.percentile(ts, 95):.4f}; back-to-back {e0.elapsed_time(e1) / n:.4f} ms per frame", flush=True)


if __name__ == "__main__":
    run(32, 512, 768, 5, 14, 4, 4)
    run(5, 64, 64, 2, 0, 0, 5)
    run(3, 32, 96, 5, 14, 4, 3, per_call=2)
    run(40, 64, 128, 3, 4, 2, 3, per_call=3)
    run(17, 512, 512, 5, 0, 0, 3, per_call=4)
    timing()
