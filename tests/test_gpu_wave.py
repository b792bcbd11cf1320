"""Round 6: the step-granular layer wavefront (csrc/gru_wave.hip; b2t_gru_wave_fwd_f32 / b2t_gru_wave_bwd_f32) -- the whole GRU
stack's sweeps as ONE launch per direction, replacing nn.GRU(num_layers = L) (model_training/rnn_model.py:65-72,126) under
torch.autocast(bfloat16) (rnn_trainer.py:527) and its autograd backward.  Through the C ABI against the same recurrences in numpy
with every matrix-core operand rounded to bf16 explicitly (the contract of the bf16 mode: tests/test_gpu_parity.py
test_persistent_sweep_bf16_operands), layer by layer, dropout masks from b2t_dropout_mask_f32."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nejm-brain-to-text_amd"))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


def _bf16_round(x):
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).bfloat16().float().numpy()


def _masks(L, T, B, H, p, seeds, elem0):
    """The keep / scale factors b2t_dropout_f32 applies to out[l] (flat element (t, row, unit), offset elem0)."""
    import b2t_native as N, b2t_ops as ops
    out = []
    for l in range(L - 1):
        m = torch.empty(T * B * H, device=_dev())
        N.check(N.load().b2t_dropout_mask_f32(ops._p(m), T * B * H, float(p), C.c_uint64(seeds[l]), elem0, ops._stream()), "mask")
        out.append(m.view(T, B, H).cpu().numpy().astype(np.float64))
    return out


def _reference(gi0, whh, bhh, wih, bih, h0, masks, dY, dhl):
    """fp64 recurrences with bf16-rounded matrix operands.  Returns outs, outd, reserves, dG, dh_init."""
    L = len(whh)
    T, B, _ = gi0.shape
    H = whh[0].shape[1]
    sig = lambda v: 1.0 / (1.0 + np.exp(-v))
    outs, outd, res = [], [], []
    x = None
    for l in range(L):
        wq = _bf16_round(whh[l]).astype(np.float64)
        gi = gi0.astype(np.float64) if l == 0 else _bf16_round(x.reshape(T * B, H)).astype(np.float64).reshape(T, B, H) @ _bf16_round(wih[l]).astype(np.float64).T + bih[l]
        h = h0[l].astype(np.float64)
        o, rs = [], []
        for t in range(T):
            gh = _bf16_round(h).astype(np.float64) @ wq.T + bhh[l]
            r = sig(gi[t][:, :H] + gh[:, :H]); z = sig(gi[t][:, H:2 * H] + gh[:, H:2 * H])
            n = np.tanh(gi[t][:, 2 * H:] + r * gh[:, 2 * H:])
            hp = h
            h = (1 - z) * n + z * hp
            o.append(h); rs.append((r, z, n, gh[:, 2 * H:], hp))
        o = np.stack(o)
        outs.append(o); res.append(rs)
        x = o * masks[l] if (masks and l < L - 1) else o
        outd.append(x)
    dG, dh_init = [None] * L, [None] * L
    dy = dY.astype(np.float64)
    for l in range(L - 1, -1, -1):
        wq = _bf16_round(whh[l]).astype(np.float64)
        carry = dhl[l].astype(np.float64) if dhl is not None else np.zeros((B, H))
        g = np.zeros((T, B, 4 * H))
        for t in range(T - 1, -1, -1):
            r, z, n, ghn, hp = res[l][t]
            d = dy[t] + carry
            dn = d * (1 - z); dz = d * (hp - n)
            dn_pre = dn * (1 - n * n); dz_pre = dz * z * (1 - z); dr_pre = dn_pre * ghn * r * (1 - r)
            g[t] = np.concatenate([dr_pre, dz_pre, dn_pre * r, dn_pre], axis=1)
            carry = d * z + _bf16_round(g[t][:, :3 * H]).astype(np.float64) @ wq
        dG[l], dh_init[l] = g, carry
        if l > 0:
            dgi = np.concatenate([g[:, :, :2 * H], g[:, :, 3 * H:]], axis=2).reshape(T * B, 3 * H)
            dy = (_bf16_round(dgi).astype(np.float64) @ _bf16_round(wih[l]).astype(np.float64)).reshape(T, B, H)
            if masks:
                dy = dy * masks[l - 1]
    return outs, outd, res, dG, dh_init


def _run(L, T, B, H, p, seed, with_dhl=True, elem0=0, reference=True):
    import b2t_native as N, b2t_ops as ops
    lib, dev, P = N.load(), _dev(), ops._p
    assert lib.b2t_gru_wave_supported(L, T, B, H) >= 1
    g = torch.Generator().manual_seed(seed)
    rnd = lambda *s, sc=1.0: torch.randn(*s, generator=g) * sc
    gi0 = rnd(T, B, 3 * H, sc=0.5)
    whh = [rnd(3 * H, H, sc=1.0 / H ** 0.5) for _ in range(L)]
    wih = [None] + [rnd(3 * H, H, sc=1.0 / H ** 0.5) for _ in range(L - 1)]
    bhh = [rnd(3 * H, sc=0.1) for _ in range(L)]
    bih = [None] + [rnd(3 * H, sc=0.1) for _ in range(L - 1)]
    h0 = [rnd(B, H, sc=0.3) for _ in range(L)]
    dY = rnd(T, B, H, sc=0.05)
    dhl = rnd(L, B, H, sc=0.05) if with_dhl else None
    seeds = [1234567 + 101 * l for l in range(L)]
    masks = _masks(L, T, B, H, p, seeds, elem0) if p > 0 and reference else None
    npf = lambda x: None if x is None else x.numpy()
    ref = None if not reference else _reference(gi0.numpy(), [w.numpy() for w in whh], [b.numpy() for b in bhh], [npf(w) for w in wih], [npf(b) for b in bih],
                     [h.numpy() for h in h0], masks, dY.numpy(), npf(dhl))

    to = lambda x: None if x is None else x.to(dev).contiguous()
    d_gi0, d_whh, d_wih, d_bhh, d_bih, d_h0, d_dY, d_dhl = to(gi0), [to(w) for w in whh], [to(w) for w in wih], [to(b) for b in bhh], [to(b) for b in bih], [to(h) for h in h0], to(dY), to(dhl)
    out = [torch.full((T, B, H), float("nan"), device=dev) for _ in range(L)]
    outd = [torch.full((T, B, H), float("nan"), device=dev) for _ in range(L)]
    res = [torch.full((T, B, 4 * H), float("nan"), device=dev) for _ in range(L)]
    dG = [torch.full((T, B, 4 * H), float("nan"), device=dev) for _ in range(L)]
    dh_init = torch.full((L, B, H), float("nan"), device=dev)
    whh_t = [w.t().contiguous() for w in d_whh]
    wih_t = [None] + [w.t().contiguous() for w in d_wih[1:]]
    err = torch.zeros(16 + 16 * 8, dtype=torch.int32, device=dev)
    d = N.WaveDesc()
    d.L, d.T, d.B, d.H = L, T, B, H
    d.gi0 = d_gi0.data_ptr()
    for l in range(L):
        d.w_hh[l], d.b_hh[l] = d_whh[l].data_ptr(), d_bhh[l].data_ptr()
        if l > 0:
            d.w_ih[l], d.b_ih[l], d.w_ih_t[l] = d_wih[l].data_ptr(), d_bih[l].data_ptr(), wih_t[l].data_ptr()
        d.h_init[l], d.out[l], d.outd[l], d.reserve[l] = d_h0[l].data_ptr(), out[l].data_ptr(), outd[l].data_ptr(), res[l].data_ptr()
        d.w_hh_t[l], d.dG[l] = whh_t[l].data_ptr(), dG[l].data_ptr()
        d.seed[l] = seeds[l]
    d.dY_top, d.dh_last, d.dh_init = d_dY.data_ptr(), (d_dhl.data_ptr() if d_dhl is not None else None), dh_init.data_ptr()
    d.drop_p, d.elem0 = float(p), elem0
    # garbage (NaN patterns) in the workspace: the rings must not need initialising
    wsf = torch.full((lib.b2t_gru_wave_ws_bytes(L, T, B, H, 0, int(p > 0)) // 4 + 64,), float("nan"), device=dev)
    wsb = torch.full((lib.b2t_gru_wave_ws_bytes(L, T, B, H, 1, int(p > 0)) // 4 + 64,), float("nan"), device=dev)
    N.check(lib.b2t_gru_wave_fwd_f32(C.byref(d), P(wsf), P(err), ops._stream()), "wave fwd")
    N.check(lib.b2t_gru_wave_bwd_f32(C.byref(d), P(wsb), P(err), ops._stream()), "wave bwd")
    torch.cuda.synchronize()
    assert int(err[0]) == 0, "hand-off timeout"
    return ref, dict(out=out, outd=outd, res=res, dG=dG, dh_init=dh_init, masks=masks, desc=d, keep=(d_gi0, d_whh, d_wih, d_bhh, d_bih, d_h0, d_dY, d_dhl, whh_t, wih_t, wsf, wsb, err))


@pytest.mark.parametrize("L,T,B,H,p", [(1, 5, 16, 32, 0.0), (2, 7, 5, 48, 0.0), (3, 9, 17, 80, 0.0), (3, 9, 17, 80, 0.3), (5, 12, 64, 128, 0.4),
                                       (5, 10, 64, 512, 0.0), (5, 6, 40, 768, 0.4), (2, 40, 33, 256, 0.2),
                                       # the K-split form (H % 128 == 0, local placement): fewer steps than its ring is deep, one / two / three
                                       # row groups (a workgroup's second row group missing), every K-quarter length
                                       (3, 3, 20, 384, 0.3), (4, 30, 16, 512, 0.0), (5, 25, 64, 512, 0.4), (2, 19, 48, 128, 0.0),
                                       (2, 150, 64, 512, 0.0), (1, 200, 40, 256, 0.0)])
def test_wavefront_against_the_bf16_operand_recurrences(L, T, B, H, p):
    _check_wavefront(L, T, B, H, p)


@pytest.mark.parametrize("L,T,B,H,p", [(5, 10, 64, 512, 0.4), (2, 40, 33, 256, 0.2)])
def test_wavefront_16_unit_form_where_the_k_split_form_would_run(monkeypatch, L, T, B, H, p):
    monkeypatch.setenv("B2T_WAVE_KS", "0")       # (read per call)
    _check_wavefront(L, T, B, H, p)


def _check_wavefront(L, T, B, H, p):
    (outs, outd, res, dG_ref, dh_ref), got = _run(L, T, B, H, p, seed=L * 1000 + H + B)
    for l in range(L):
        o = got["out"][l].cpu().numpy()
        assert np.isfinite(o).all(), f"layer {l}: non-finite outputs"
        # rounding decisions can flip where fp32 and fp64 intermediates straddle a bf16 boundary: compare at bf16-ulp scale
        # (a flipped rounding in layer l moves the inputs of every layer above it: the bound grows with the layer)
        tol = 3e-3 * (1 + l)
        np.testing.assert_allclose(o, outs[l], atol=tol, err_msg=f"out[{l}]")
        r_ref = np.stack([np.concatenate(rs[:4], axis=1) for rs in res[l]])
        np.testing.assert_allclose(got["res"][l].cpu().numpy(), r_ref, atol=tol, err_msg=f"reserve[{l}]")
        if p > 0 and l < L - 1:
            # the dropped copy is EXACTLY mask x the kernel's own output (same Philox draws as b2t_dropout_f32)
            want = (got["out"][l].cpu().numpy().astype(np.float32) * got["masks"][l].astype(np.float32))
            np.testing.assert_array_equal(got["outd"][l].cpu().numpy(), want)
        sc = max(1.0, float(np.abs(dG_ref[l]).max()))
        np.testing.assert_allclose(got["dG"][l].cpu().numpy(), dG_ref[l], atol=3e-3 * L * sc, err_msg=f"dG[{l}]")
        np.testing.assert_allclose(got["dh_init"][l].cpu().numpy(), dh_ref[l], atol=3e-3 * L * max(1.0, float(np.abs(dh_ref[l]).max())), err_msg=f"dh_init[{l}]")


@pytest.mark.parametrize("L,T,B,H,p", [(4, 24, 50, 96, 0.25), (5, 60, 64, 512, 0.4), (1, 120, 64, 512, 0.0), (3, 90, 33, 256, 0.0)])
def test_wavefront_is_repeatable_and_needs_no_clean_workspace(L, T, B, H, p):
    """Two calls on the same (dirty) workspace give bit-identical results: nothing depends on the order in which the workgroups
    of a step arrive, and the call clears its own counters (K-split form: re-arms its own ring)."""
    import b2t_native as N, b2t_ops as ops
    lib, P = N.load(), ops._p
    _, got = _run(L, T, B, H, p, seed=5, reference=False)
    first = [t.clone() for t in got["out"] + got["dG"]] + [got["dh_init"].clone()]
    assert all(bool(torch.isfinite(t).all()) for t in first)
    d, keep = got["desc"], got["keep"]
    wsf, wsb, err = keep[-3], keep[-2], keep[-1]
    for _ in range(3):
        N.check(lib.b2t_gru_wave_fwd_f32(C.byref(d), P(wsf), P(err), ops._stream()), "wave fwd")
        N.check(lib.b2t_gru_wave_bwd_f32(C.byref(d), P(wsb), P(err), ops._stream()), "wave bwd")
        torch.cuda.synchronize()
        assert int(err[0]) == 0
        again = got["out"] + got["dG"] + [got["dh_init"]]
        for a, b in zip(first, again):
            assert torch.equal(a, b)


def test_wavefront_refuses_what_it_cannot_hold():
    import b2t_native as N
    lib = N.load()
    assert lib.b2t_gru_wave_supported(5, 100, 64, 768) == 1          # 240 workgroups
    assert lib.b2t_gru_wave_supported(5, 100, 64, 1024) == 0         # H > 768
    assert lib.b2t_gru_wave_supported(8, 100, 64, 768) == 0          # 384 workgroups > CUs
    assert lib.b2t_gru_wave_supported(5, 100, 65, 512) == 0          # B > 64
    assert lib.b2t_gru_wave_supported(5, 100, 64, 520) == 0          # H % 16
    # the form: K-split where H % 128 == 0, H <= 512 and a layer's workgroups fit one XCD (8 XCDs x 32 CUs, round-robin dispatch verified)
    if torch.cuda.get_device_properties(0).multi_processor_count >= 256:
        assert lib.b2t_gru_wave_supported(5, 100, 64, 512) == 2
        assert lib.b2t_gru_wave_supported(5, 100, 64, 384) == 2
    assert lib.b2t_gru_wave_supported(5, 100, 64, 320) == 1          # H % 128
    assert lib.b2t_gru_wave_supported(5, 5000, 64, 512) == 1         # saved gates >= 2 GB: 32-bit offsets of the K-split form


def _step_args():
    return dict(lr_max=0.005, lr_min=0.0001, lr_decay_steps=120000, lr_warmup_steps=1000, lr_max_day=0.005, lr_min_day=0.0001,
                lr_decay_steps_day=120000, lr_warmup_steps_day=1000, beta0=0.9, beta1=0.999, epsilon=0.1, weight_decay=0.001,
                weight_decay_day=0, grad_norm_clip_value=10)


@pytest.mark.parametrize("F,H,D,Cc,L,B,T,S,patch,drop", [(128, 256, 4, 41, 3, 48, 230, 8, (6, 3), (0.3, 0.2)),
                                                        (256, 256, 3, 41, 2, 32, 200, 8, (0, 0), (0.0, 0.0)),
                                                        (64, 96, 3, 41, 5, 20, 90, 6, (0, 0), (0.0, 0.4))])
def test_bf16_mode_step_on_the_wavefront_tracks_the_chunk_pipeline(monkeypatch, F, H, D, Cc, L, B, T, S, patch, drop):
    """The training step of the bf16 mode (`use_amp`) with its sweeps as the layer wavefront (default) against the round-5 chunk
    pipeline (B2T_WAVE=0): the same operands rounded to bf16 at the same places, the same dropout masks (Philox stream of
    b2t_dropout_f32 on the layers' outputs), fp32 accumulation in another order -- logits, losses and every gradient agree at the
    bf16 mode's own scale (a rounding that flips between the two orders moves an operand by 2^-9)."""
    import b2t_ops as ops
    from rnn_model import GRUDecoder
    from b2t_train_step import TrainStep
    dev = _dev()
    old_amp = ops.AMP["on"]
    try:
        ops.set_amp(True)
        g = torch.Generator().manual_seed(B + H + T)
        x = torch.randn(B, T, F, generator=g).to(dev)
        day = torch.randint(0, D, (B,), generator=g)
        tgt = torch.randint(1, Cc, (B, S), generator=g); tl = torch.randint(2, S + 1, (B,), generator=g)
        nt = torch.randint(T - 10, T + 1, (B,), generator=g)
        for b in range(B):
            tgt[b, tl[b]:] = 0

        def grads(wave, chunks="1,1"):
            monkeypatch.setenv("B2T_WAVE", "1" if wave else "0")
            monkeypatch.setenv("B2T_WAVE_CHUNKS", chunks)
            monkeypatch.setenv("B2T_WAVE_DIRS", "fb")      # both passes on the wavefront (the trainer's default is the forward pass only)
            torch.manual_seed(3)
            m = GRUDecoder(F, H, D, Cc, drop[1], drop[0], L, patch[0], patch[1]).to(dev).train()
            ts = TrainStep(m, _step_args())
            loss_b = ts.compute_grads(x, day, tgt, nt, tl)
            torch.cuda.synchronize()
            ts.check_status()
            return ts.grad_arena.clone(), loss_b.clone(), ts.last_logits.clone()

        ref, got, again = grads(False), grads(True), grads(True)
        assert torch.isfinite(ref[0]).all() and float(ref[0].abs().max()) > 0
        for a, r, name in zip(got, ref, ("gradients", "losses", "logits")):
            scale = float(r.abs().max())
            diff = float((a - r).abs().max())
            print(f"[wave vs chunks] H={H} L={L} {name}: max diff {diff:.3e} of scale {scale:.3e}")
            assert diff <= 5e-3 * scale, f"{name} differ by {diff} (scale {scale})"
        for a, b in zip(got, again):       # the wavefront itself is deterministic
            assert torch.equal(a, b)
        # forward: a launch per time chunk (states carried in fp32 exactly as inside one launch); backward: the consumers of the sweep's
        # dG -- layer 0's input gradient, every weight gradient -- chunk by chunk BESIDE the one sweep launch, each behind a gate on the
        # sweep's progress word: logits and losses bit-identical, the weight gradients accumulated chunk by chunk (another fp32
        # summation order)
        cut = grads(True, "3,2")
        assert torch.equal(cut[2], got[2]) and torch.equal(cut[1], got[1])
        assert float((cut[0] - got[0]).abs().max()) <= 2e-4 * float(got[0].abs().max())
        # ... and the default: forward pass on the wavefront, backward pass on the chunk pipeline
        monkeypatch.setenv("B2T_WAVE_DIRS", "f")
        torch.manual_seed(3)
        m = GRUDecoder(F, H, D, Cc, drop[1], drop[0], L, patch[0], patch[1]).to(dev).train()
        ts = TrainStep(m, _step_args())
        loss_b = ts.compute_grads(x, day, tgt, nt, tl)
        torch.cuda.synchronize(); ts.check_status()
        assert torch.equal(ts.last_logits, got[2]) and torch.equal(loss_b, got[1])
        assert float((ts.grad_arena - ref[0]).abs().max()) <= 5e-3 * float(ref[0].abs().max())
        # ... and it is the wavefront that ran: the two are not bit-identical
        assert not torch.equal(got[2], ref[2])
    finally:
        ops.set_amp(old_amp)
        monkeypatch.delenv("B2T_WAVE", raising=False)
        monkeypatch.delenv("B2T_WAVE_CHUNKS", raising=False)
        monkeypatch.delenv("B2T_WAVE_DIRS", raising=False)
