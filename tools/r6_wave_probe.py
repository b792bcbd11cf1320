#!/usr/bin/env python3
"""Round 6: launch times of the layer wavefront (csrc/gru_wave.hip) alone on the chip, at the bench shapes: us per launch and per
time step of the chain (T + L - 1 steps), forward and backward, with and without the inter-layer dropout; L = 1 for the bare chain."""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "nejm-brain-to-text_amd"))
import torch
import b2t_native as N, b2t_ops as ops
lib, dev, P = N.load(), torch.device("cuda:0"), ops._p

def probe(L, T, B, H, p, reps=5, seed=1, ws=None, timing=True):
    g = torch.Generator().manual_seed(seed)
    rnd = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
    gi0 = rnd(T, B, 3 * H, sc=0.5)
    whh = [rnd(3 * H, H, sc=1.0 / H ** 0.5) for _ in range(L)]; wih = [rnd(3 * H, H, sc=1.0 / H ** 0.5) for _ in range(L)]
    bhh = [rnd(3 * H, sc=0.1) for _ in range(L)]; bih = [rnd(3 * H, sc=0.1) for _ in range(L)]
    h0 = [rnd(B, H, sc=0.3) for _ in range(L)]
    dY = rnd(T, B, H, sc=0.05)
    out = [torch.empty(T, B, H, device=dev) for _ in range(L)]; outd = [torch.empty(T, B, H, device=dev) for _ in range(L)]
    res = [torch.empty(T, B, 4 * H, device=dev) for _ in range(L)]; dG = [torch.empty(T, B, 4 * H, device=dev) for _ in range(L)]
    dh = torch.empty(L, B, H, device=dev)
    whh_t = [w.t().contiguous() for w in whh]; wih_t = [w.t().contiguous() for w in wih]
    err = torch.zeros(16 + 16 * 8, dtype=torch.int32, device=dev)
    d = N.WaveDesc(); d.L, d.T, d.B, d.H = L, T, B, H; d.gi0 = gi0.data_ptr()
    for l in range(L):
        d.w_hh[l], d.b_hh[l], d.w_ih[l], d.b_ih[l], d.w_ih_t[l] = whh[l].data_ptr(), bhh[l].data_ptr(), wih[l].data_ptr(), bih[l].data_ptr(), wih_t[l].data_ptr()
        d.h_init[l], d.out[l], d.outd[l], d.reserve[l] = h0[l].data_ptr(), out[l].data_ptr(), outd[l].data_ptr(), res[l].data_ptr()
        d.w_hh_t[l], d.dG[l], d.seed[l] = whh_t[l].data_ptr(), dG[l].data_ptr(), 77 + l
    d.dY_top, d.dh_last, d.dh_init, d.drop_p, d.elem0 = dY.data_ptr(), None, dh.data_ptr(), float(p), 0
    if ws is None:
        ws = (torch.empty(lib.b2t_gru_wave_ws_bytes(L, T, B, H, 0, int(p > 0)) // 4 + 64, device=dev),
              torch.empty(lib.b2t_gru_wave_ws_bytes(L, T, B, H, 1, int(p > 0)) // 4 + 64, device=dev))
    wsf, wsb = ws
    r = dict(L=L, T=T, B=B, H=H, p=p, sc1_loads=os.environ.get("B2T_WAVE_SC1_LOADS", "0"), local=os.environ.get("B2T_WAVE_LOCAL", "1"))
    if not timing:
        N.check(lib.b2t_gru_wave_fwd_f32(C.byref(d), P(wsf), P(err), ops._stream()), "fwd")
        N.check(lib.b2t_gru_wave_bwd_f32(C.byref(d), P(wsb), P(err), ops._stream()), "bwd")
        torch.cuda.synchronize()
        assert int(err[0]) == 0, "hand-off timeout"
        return [o.clone() for o in out + dG] + [dh.clone()], ws
    for name, fn, ws in (("fwd", lib.b2t_gru_wave_fwd_f32, wsf), ("bwd", lib.b2t_gru_wave_bwd_f32, wsb)):
        ts = []
        for i in range(reps + 1):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); N.check(fn(C.byref(d), P(ws), P(err), ops._stream()), name); e1.record(); torch.cuda.synchronize()
            if i: ts.append(e0.elapsed_time(e1) * 1e3)
        assert int(err[0]) == 0, "hand-off timeout"
        r[name + "_us"] = round(min(ts), 1); r[name + "_us_per_step"] = round(min(ts) / (T + L - 1), 3); r[name + "_all"] = [round(t, 1) for t in ts]
        if os.environ.get("B2T_LIB", ""):     # cycles per step: wait own, loads + product, gates + tile, drain + counters, stores, wait neighbour, projection
            tm = err[16 + (0 if name == "fwd" else 64):][:8 * L].view(L, 8).cpu().numpy()
            r[name + "_cycles_per_step_by_layer"] = [[int(v) for v in row[:8]] for row in tm]
    assert all(torch.isfinite(o).all() for o in out + dG)
    print("R6WAVE " + json.dumps(r), flush=True)

# stale-line hunt for the ordinary (L2-served) fragment loads: inputs A then inputs B on the SAME workspace (same ring addresses)
# must equal inputs B on a fresh workspace, bit for bit; repeated so that the tail of one pass is still in the L2s when the next starts
for cfg in (() if os.environ.get('R6_PROBE_ONLY_TIMING') else ((5, 500, 64, 512, 0.4), (5, 122, 64, 768, 0.4), (3, 40, 64, 256, 0.0))):
    ok = True
    for rep in range(3):
        _, ws = probe(*cfg, seed=10 + rep, timing=False)
        got, _ = probe(*cfg, seed=20 + rep, ws=ws, timing=False)
        want, _ = probe(*cfg, seed=20 + rep, timing=False)
        ok = ok and all(torch.equal(a, b) for a, b in zip(got, want))
    print("R6STALE " + json.dumps(dict(cfg=cfg, sc1_loads=os.environ.get("B2T_WAVE_SC1_LOADS", "0"), local=os.environ.get("B2T_WAVE_LOCAL", "1"), second_pass_on_used_workspace_equals_fresh=ok)), flush=True)
CFGS = ((5, 500, 64, 512, 0.0), (5, 500, 64, 512, 0.4), (1, 500, 64, 512, 0.0), (5, 122, 64, 768, 0.4), (1, 122, 64, 768, 0.0))
if os.environ.get("R6_PROBE_CFGS"):
    CFGS = tuple(tuple(float(x) if "." in x else int(x) for x in c.split(",")) for c in os.environ["R6_PROBE_CFGS"].split(";"))
for cfg in CFGS:
    probe(*cfg)
