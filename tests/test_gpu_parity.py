"""GPU parity: the HIP path (through the C ABI) vs the oracle and the golden vectors captured from
the reference.  Tolerances (fp32 path, stated per SURVEY §8b numerics contract):
  logits / hidden  : abs 1e-4          CTC loss : rel 1e-5        dlogits : abs 5e-6
  parameter grads  : 1e-3 of the tensor's max-abs                  AdamW params after 4 steps: abs 2e-5
  greedy argmax indices : identical wherever the oracle's top-2 margin exceeds 1e-4
"""
import os

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


def make_model(cfg, sd):
    from rnn_model import GRUDecoder
    F, H, D, C, L, ps, st = [int(v) for v in cfg]
    m = GRUDecoder(F, H, D, C, 0.0, 0.0, L, ps, st)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    return m.to(_dev())


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("akc,bkc", [(1, 1), (1, 0), (0, 1), (0, 0)])
@pytest.mark.parametrize("M,N,K", [(128, 128, 16), (200, 41, 37), (33, 300, 129), (512, 1536, 512)])
def test_gemm(akc, bkc, M, N, K):
    import b2t_ops as ops
    dev = _dev()
    rng = np.random.default_rng(M * 7 + N * 3 + K + akc * 2 + bkc)
    lda = (K if akc else M) + 4
    ldb = (K if bkc else N) + 8
    lda += (-lda) % 4; ldb += (-ldb) % 4
    A = rng.standard_normal((M if akc else K, lda)).astype(np.float32)
    Bm = rng.standard_normal((N if bkc else K, ldb)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    Am = A[:, :K] if akc else A[:, :M].T            # asymmetric operands: transposes are detected
    Bn = Bm[:, :K] if bkc else Bm[:, :N].T
    ref = Am.astype(np.float64) @ Bn.astype(np.float64).T + bias
    tA, tB = torch.from_numpy(A).to(dev), torch.from_numpy(Bm).to(dev)
    tC = torch.zeros((M, N + 3), device=dev)
    ops.gemm(tA, tB, tC, M=M, N_=N, K=K, a_kc=akc, b_kc=bkc, a_s0=lda, b_s0=ldb, c_s0=N + 3,
             bias=torch.from_numpy(bias).to(dev))
    got = tC.cpu().numpy()
    np.testing.assert_allclose(got[:, :N], ref, atol=2e-5 * np.sqrt(K) * 4, rtol=1e-5)
    assert np.all(got[:, N:] == 0)                   # no out-of-bounds writes
    # accumulate + softsign epilogue
    tC2 = torch.ones((M, N), device=dev)
    ops.gemm(tA, tB, tC2, M=M, N_=N, K=K, a_kc=akc, b_kc=bkc, a_s0=lda, b_s0=ldb, c_s0=N, accumulate=1)
    np.testing.assert_allclose(tC2.cpu().numpy(), ref - bias + 1.0, atol=2e-5 * np.sqrt(K) * 4, rtol=1e-5)
    tC3 = torch.zeros((M, N), device=dev)
    ops.gemm(tA, tB, tC3, M=M, N_=N, K=K, a_kc=akc, b_kc=bkc, a_s0=lda, b_s0=ldb, c_s0=N, epilogue=1)
    r3 = ref - bias
    np.testing.assert_allclose(tC3.cpu().numpy(), r3 / (1 + np.abs(r3)), atol=2e-5 * np.sqrt(K) * 4)  # d softsign <= 1


def _bf16_round(x):
    """fp32 -> nearest-even bf16 -> fp32 (what v_cvt_pk_bf16_f32 does to the operands)."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((u >> 16) & 1) + 0x7FFF
    return ((u + r) & 0xFFFF0000).astype(np.uint32).view(np.float32)


@pytest.mark.parametrize("akc,bkc", [(1, 1), (1, 0), (0, 1), (0, 0)])
@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (200, 41, 37), (33, 300, 129), (512, 1536, 512), (64, 64, 41)])
def test_gemm_bf16(akc, bkc, M, N, K):
    """b2t_gemm_bf16_f32 (the `use_amp` matmul regime): operands rounded to bf16 (RNE), fp32 accumulate and output --
    equal to an fp64 product of the rounded operands up to fp32 summation roundoff, for all four operand layouts."""
    import b2t_native as Nn
    import b2t_ops as ops
    dev = _dev()
    rng = np.random.default_rng(M * 7 + N * 3 + K)
    A = rng.standard_normal((M, K)).astype(np.float32)
    Bm = rng.standard_normal((N, K)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    ref = _bf16_round(A).astype(np.float64) @ _bf16_round(Bm).astype(np.float64).T + bias
    Mp, Np, Kp = (M + 3) // 4 * 4, (N + 3) // 4 * 4, (K + 3) // 4 * 4
    if akc:
        Ad = torch.zeros(M, Kp); Ad[:, :K] = torch.from_numpy(A); a_s0 = Kp
    else:
        Ad = torch.zeros(K, Mp); Ad[:, :M] = torch.from_numpy(A.T); a_s0 = Mp
    if bkc:
        Bd = torch.zeros(N, Kp); Bd[:, :K] = torch.from_numpy(Bm); b_s0 = Kp
    else:
        Bd = torch.zeros(K, Np); Bd[:, :N] = torch.from_numpy(Bm.T); b_s0 = Np
    Ad, Bd = Ad.to(dev), Bd.to(dev)
    Cd = torch.full((M, N), float("nan"), device=dev)
    old = ops.AMP["on"]
    ops.set_amp(True)
    try:
        ops.gemm(Ad, Bd, Cd, M=M, N_=N, K=K, a_kc=akc, a_s0=a_s0, b_kc=bkc, b_s0=b_s0, c_s0=N, bias=torch.from_numpy(bias).to(dev))
        got = Cd.cpu().numpy()
        np.testing.assert_allclose(got, ref, atol=2e-5 * max(1.0, float(np.abs(ref).max())))
        # and it really is the bf16 product, not the fp32 one
        if K >= 32:
            exact = A.astype(np.float64) @ Bm.astype(np.float64).T + bias
            assert np.abs(got - exact).max() > 10 * np.abs(got - ref).max()
        # split-K slabs (weight-gradient shape: long K) agree with the single pass
        if M == 512:
            ws = ops.Workspace()
            C2 = torch.empty((M, N), device=dev)
            ops.gemm(Ad, Bd, C2, M=M, N_=N, K=K, a_kc=akc, a_s0=a_s0, b_kc=bkc, b_s0=b_s0, c_s0=N, splitk=4, ws=ws)
            np.testing.assert_allclose(C2.cpu().numpy() + bias, ref, atol=2e-5 * float(np.abs(ref).max()))
    finally:
        ops.set_amp(old)


@pytest.mark.parametrize("B,H,T,wide", [(17, 80, 9, 0), (64, 512, 12, 0), (40, 768, 6, 0), (64, 512, 12, 1), (23, 96, 7, 1), (40, 768, 6, 1), (64, 768, 20, 1), (64, 512, 180, 1)])
def test_persistent_sweep_bf16_operands(B, H, T, wide):
    """mode 1 | B2T_GRU_BF16: the recurrent products round their operands (h_{t-1} / dG_{t+1} and the W_hh slice) to bf16
    and accumulate in fp32.  Reference: the same recurrences in numpy with the operands rounded explicitly."""
    import b2t_native as Nn
    import b2t_ops as ops
    lib = Nn.load(); dev = _dev(); p = ops._p
    g = torch.Generator().manual_seed(B + H)
    rnd = lambda *s: torch.randn(*s, generator=g)
    gi, w, b_, h0 = rnd(T, B, 3 * H) * 0.5, rnd(3 * H, H) * (1.0 / H ** 0.5), rnd(3 * H) * 0.1, rnd(B, H) * 0.3
    dY, dhl = rnd(T, B, H) * 0.05, rnd(B, H) * 0.05
    sig = lambda v: 1.0 / (1.0 + np.exp(-v))
    wq = _bf16_round(w.numpy()).astype(np.float64)
    # forward
    h = h0.numpy().astype(np.float64)
    outs, res = [], []
    for t in range(T):
        gh = _bf16_round(h.astype(np.float32)).astype(np.float64) @ wq.T + b_.numpy()
        git = gi[t].numpy().astype(np.float64)
        r = sig(git[:, :H] + gh[:, :H]); z = sig(git[:, H:2 * H] + gh[:, H:2 * H])
        n = np.tanh(git[:, 2 * H:] + r * gh[:, 2 * H:])
        hprev = h
        h = (1 - z) * n + z * hprev
        outs.append(h); res.append((r, z, n, gh[:, 2 * H:], hprev))
    # backward, dGh rounded for the carry product
    carry = dhl.numpy().astype(np.float64)
    dG_ref = np.zeros((T, B, 4 * H))
    for t in range(T - 1, -1, -1):
        r, z, n, ghn, hprev = res[t]
        d = dY[t].numpy() + carry
        dn = d * (1 - z); dz = d * (hprev - n)
        dn_pre = dn * (1 - n * n); dz_pre = dz * z * (1 - z); dr_pre = dn_pre * ghn * r * (1 - r)
        dG_ref[t] = np.concatenate([dr_pre, dz_pre, dn_pre * r, dn_pre], axis=1)
        carry = d * z + _bf16_round(dG_ref[t][:, :3 * H].astype(np.float32)).astype(np.float64) @ wq
    gi, w, b_, h0, dY, dhl = (x.to(dev) for x in (gi, w, b_, h0, dY, dhl))
    wt = w.t().contiguous()
    out = torch.zeros(T + 1, B, H, device=dev); out[0] = h0
    resv = torch.zeros(T, B, 4 * H, device=dev)
    sync = torch.zeros(lib.b2t_gru_ws_bytes(T, B, H) // 4 + 16, dtype=torch.int32, device=dev)
    Nn.check(lib.b2t_gru_layer_fwd_f32(p(gi), p(w), p(b_), p(out[0]), p(out[1:]), p(resv), None, T, B, H, 1 | ops.GRU_BF16 | (ops.GRU_WIDE if wide else 0),
                                       p(sync), ops._stream()), "fwd")
    dG = torch.zeros(T, B, 4 * H, device=dev); dh = torch.zeros(B, H, device=dev); sc = torch.empty(B, H, device=dev)
    Nn.check(lib.b2t_gru_layer_bwd_f32(p(dY), p(dhl), p(resv), p(out[1:]), p(out[0]), p(wt), p(dG), p(dh), p(sc), T, B, H,
                                       1 | ops.GRU_BF16 | (ops.GRU_WIDE if wide else 0), p(sync), ops._stream()), "bwd")
    torch.cuda.synchronize()
    assert int(sync[0]) == 0
    # rounding decisions can flip where fp32 and fp64 intermediates straddle a bf16 boundary: compare at bf16-ulp scale
    np.testing.assert_allclose(out[1:].cpu().numpy(), np.stack(outs), atol=2e-3)
    np.testing.assert_allclose(dG.cpu().numpy(), dG_ref, atol=2e-3 * max(1.0, float(np.abs(dG_ref).max())))
    np.testing.assert_allclose(dh.cpu().numpy(), carry, atol=2e-3 * max(1.0, float(np.abs(carry).max())))
    # and it is not the fp32 path
    out32 = torch.zeros(T + 1, B, H, device=dev); out32[0] = h0
    Nn.check(lib.b2t_gru_layer_fwd_f32(p(gi), p(w), p(b_), p(out32[0]), p(out32[1:]), None, None, T, B, H, 1, p(sync), ops._stream()), "fwd32")
    assert float((out32 - out).abs().max()) > 1e-5
    if wide:
        # The 32-unit workgroups hand their tiles over as bf16 MFMA fragments through the buffer behind the counters
        # (gru_sync.h); with B2T_HANDOFF16=0 as fp32 tiles that every consumer transposes and rounds itself: the same roundings of
        # the same numbers, the same split of the contraction over the waves -- bit-identical results.
        import os
        old_env = os.environ.get("B2T_HANDOFF16")
        os.environ["B2T_HANDOFF16"] = "0"
        try:
            out_n = torch.zeros(T + 1, B, H, device=dev); out_n[0] = h0
            resv_n = torch.zeros(T, B, 4 * H, device=dev)
            Nn.check(lib.b2t_gru_layer_fwd_f32(p(gi), p(w), p(b_), p(out_n[0]), p(out_n[1:]), p(resv_n), None, T, B, H,
                                               1 | ops.GRU_BF16 | ops.GRU_WIDE, p(sync), ops._stream()), "fwd fp32 tiles")
            dG_n = torch.zeros(T, B, 4 * H, device=dev); dh_n = torch.zeros(B, H, device=dev)
            Nn.check(lib.b2t_gru_layer_bwd_f32(p(dY), p(dhl), p(resv_n), p(out_n[1:]), p(out_n[0]), p(wt), p(dG_n), p(dh_n), p(sc), T, B, H,
                                               1 | ops.GRU_BF16 | ops.GRU_WIDE, p(sync), ops._stream()), "bwd fp32 tiles")
            torch.cuda.synchronize()
        finally:
            if old_env is None: os.environ.pop("B2T_HANDOFF16", None)
            else: os.environ["B2T_HANDOFF16"] = old_env
        assert int(sync[0]) == 0
        assert torch.equal(out, out_n) and torch.equal(resv, resv_n)
        assert torch.equal(dG, dG_n) and torch.equal(dh, dh_n)
        # ... and under the XCD-local hand-off (ordinary loads of the fragments; both layer parities).  (64, 512, 180): the backward
        # call's 180 steps of 192 KB exceed the buffer, its slots wrap around after 170.
        for extra in (ops.GRU_LOCAL, ops.GRU_LOCAL | ops.GRU_PARITY):
            out_l = torch.zeros(T + 1, B, H, device=dev); out_l[0] = h0
            resv_l = torch.zeros(T, B, 4 * H, device=dev)
            Nn.check(lib.b2t_gru_layer_fwd_f32(p(gi), p(w), p(b_), p(out_l[0]), p(out_l[1:]), p(resv_l), None, T, B, H,
                                               1 | ops.GRU_BF16 | ops.GRU_WIDE | extra, p(sync), ops._stream()), "fwd local")
            dG_l = torch.zeros(T, B, 4 * H, device=dev); dh_l = torch.zeros(B, H, device=dev)
            Nn.check(lib.b2t_gru_layer_bwd_f32(p(dY), p(dhl), p(resv_l), p(out_l[1:]), p(out_l[0]), p(wt), p(dG_l), p(dh_l), p(sc), T, B, H,
                                               1 | ops.GRU_BF16 | ops.GRU_WIDE | extra, p(sync), ops._stream()), "bwd local")
            torch.cuda.synchronize()
            assert int(sync[0]) == 0
            assert torch.equal(out, out_l) and torch.equal(dG, dG_l) and torch.equal(dh, dh_l)


def test_train_step_bf16_matmuls_track_fp32():
    """B2T_AMP regime end to end (bf16 matmul operands everywhere, fp32 sweeps / CTC / optimizer): loss and gradients
    stay within bf16 distance of the fp32 step on the same batch, and a few steps of training reduce the loss."""
    import b2t_ops as ops
    from rnn_model import GRUDecoder
    from b2t_train_step import TrainStep
    dev = _dev()
    F, H, D, C, L, B, T = 128, 256, 3, 41, 3, 32, 48
    args = dict(lr_max=0.01, lr_min=0.001, lr_decay_steps=100, lr_warmup_steps=0, lr_scheduler_type="cosine",
                lr_max_day=0.01, lr_min_day=0.001, lr_decay_steps_day=100, lr_warmup_steps_day=0, beta0=0.9,
                beta1=0.999, epsilon=0.1, weight_decay=0.0, weight_decay_day=0, grad_norm_clip_value=10,
                _debug_keep_unclipped=True)
    torch.manual_seed(5)
    x = torch.randn(B, T, F, device=dev) * 0.5; day = torch.randint(0, D, (B,), device=dev, dtype=torch.int32)
    tgt = torch.randint(1, C, (B, 6), device=dev, dtype=torch.int32)
    nt = torch.full((B,), T, device=dev, dtype=torch.int32); tl = torch.full((B,), 6, device=dev, dtype=torch.int32)
    old = ops.AMP["on"]
    out = {}
    try:
        for amp in (False, True):
            ops.set_amp(amp)
            torch.manual_seed(6)
            model = GRUDecoder(F, H, D, C, 0.0, 0.0, L, 0, 0).to(dev).train()
            ts = TrainStep(model, dict(args))
            losses = []
            for it in range(6):
                loss, _ = ts.step(x, day, tgt, nt, tl)
                losses.append(float(loss))
                if it == 0:
                    g0 = {k: v.copy() for k, v in ts.last_unclipped_grads().items()}
            out[amp] = (losses, g0)
    finally:
        ops.set_amp(old)
    (l32, g32), (l16, g16) = out[False], out[True]
    assert abs(l16[0] - l32[0]) < 2e-2 * abs(l32[0]) and l16[0] != l32[0]
    assert l16[-1] < 0.9 * l16[0]
    for k, ref in g32.items():
        scale = max(1e-6, float(np.abs(ref).max()))
        assert float(np.abs(g16[k] - ref).max()) < 0.08 * scale, k


def test_smooth_golden(golden_dir):
    from data_augmentations import gauss_smooth
    z = load(golden_dir, "smooth.npz")
    dev = _dev()
    x = torch.from_numpy(z["x"]).to(dev)
    np.testing.assert_allclose(gauss_smooth(x, dev, 2, 100, "same").cpu().numpy(), z["same"], atol=2e-6)
    np.testing.assert_allclose(gauss_smooth(x, dev, 2, 100, "valid").cpu().numpy(), z["valid"], atol=2e-6)
    x2 = torch.from_numpy(z["x2"]).to(dev)
    np.testing.assert_allclose(gauss_smooth(x2, dev, 1, 50, "same").cpu().numpy(), z["same_std1"], atol=2e-6)


def test_transform_golden(golden_dir):
    """transform_data with the reference's noise draws injected (rnn_trainer.py:436-484)."""
    import b2t_ops as ops
    z = load(golden_dir, "transform.npz")
    dev = _dev()
    x = torch.from_numpy(z["x"]).to(dev)
    wn = torch.from_numpy(z["white"]).to(dev)
    on = torch.from_numpy(z["offset"].reshape(z["offset"].shape[0], -1)).to(dev)
    for cut in (0, 1, 2):
        y = ops.augment_smooth(x, 2, 100, "same", cut=cut, white_std=1.0, offset_std=0.2, white_noise=wn,
                               offset_noise=on)
        np.testing.assert_allclose(y.cpu().numpy(), z[f"train_cut{cut}"], atol=3e-6)
    y = ops.augment_smooth(x, 2, 100, "same")
    np.testing.assert_allclose(y.cpu().numpy(), z["val"], atol=3e-6)


def test_augment_noise_statistics():
    """Philox path: deterministic under seed, N(0, ws^2 + os^2) before smoothing, offset constant along T."""
    import b2t_ops as ops
    dev = _dev()
    x = torch.zeros((8, 256, 512), device=dev)
    a = ops.augment_smooth(x, 2, 100, "same", white_std=1.0, offset_std=0.2, seed=1234, smooth=False)
    b = ops.augment_smooth(x, 2, 100, "same", white_std=1.0, offset_std=0.2, seed=1234, smooth=False)
    c = ops.augment_smooth(x, 2, 100, "same", white_std=1.0, offset_std=0.2, seed=1235, smooth=False)
    assert torch.equal(a, b) and not torch.equal(a, c)
    an = a.cpu().numpy().astype(np.float64)
    assert abs(an.mean()) < 5e-3
    assert abs(an.var() - (1.0 + 0.04)) < 1e-2
    off = an.mean(axis=1)                     # [B,F] ~ offset draw (+ white mean / sqrt(T))
    assert abs(off.var() - (0.04 + 1.0 / 256)) < 4e-3
    only_off = ops.augment_smooth(x, 2, 100, "same", white_std=0.0, offset_std=0.2, seed=7, smooth=False).cpu().numpy()
    assert np.all(only_off == only_off[:, :1, :])
    # independence across neighbouring elements
    w = ops.augment_smooth(x, 2, 100, "same", white_std=1.0, seed=99, smooth=False).cpu().numpy().astype(np.float64)
    assert abs(np.mean(w[:, :, :-1] * w[:, :, 1:])) < 5e-3 and abs(np.mean(w[:, :-1] * w[:, 1:])) < 5e-3


@pytest.mark.parametrize("tag", ["h64", "h512", "patch", "f512"])
def test_model_forward_golden(golden_dir, tag):
    z = load(golden_dir, f"fwd_{tag}.npz")
    m = make_model(z["cfg"], sd_of(z)).eval()
    dev = _dev()
    x = torch.from_numpy(z["x"]).to(dev)
    day = torch.from_numpy(z["day_idx"]).to(dev)
    with torch.no_grad():
        logits, hidden = m(x, day, None, True)
    lg = logits.cpu().numpy()
    np.testing.assert_allclose(lg, z["logits"], atol=1e-4)
    np.testing.assert_allclose(hidden.cpu().numpy(), z["hidden"], atol=1e-4)
    # greedy argmax: identical wherever the reference's top-2 margin is > 1e-4
    srt = np.sort(z["logits"], axis=-1)
    safe = (srt[..., -1] - srt[..., -2]) > 1e-4
    assert np.array_equal(np.argmax(lg, -1)[safe], np.argmax(z["logits"], -1)[safe])
    if "stream_split" in z.files:      # carried state == whole sequence (rnn_model.py:131-132)
        t1 = int(z["stream_split"])
        with torch.no_grad():
            l1, s1 = m(x[:, :t1].contiguous(), day, None, True)
            l2, s2 = m(x[:, t1:].contiguous(), day, s1, True)
        np.testing.assert_allclose(torch.cat([l1, l2], 1).cpu().numpy(), z["stream_logits"], atol=1e-4)
        np.testing.assert_allclose(s2.cpu().numpy(), z["stream_state"], atol=1e-4)


def test_ctc_golden(golden_dir):
    import b2t_ops as ops
    z = load(golden_dir, "ctc.npz")
    dev = _dev()
    B = z["logits"].shape[0]
    loss, dl, ldd = ops.ctc_loss(torch.from_numpy(z["logits"]).to(dev), torch.from_numpy(z["targets"]),
                                 torch.from_numpy(z["in_len"]), torch.from_numpy(z["tgt_len"]), True, 1.0 / B,
                                 ops.Workspace())
    np.testing.assert_allclose(loss.cpu().numpy(), z["loss"], rtol=1e-5)
    d = dl.cpu().numpy()
    np.testing.assert_allclose(d[:, :, :41], z["dlogits"], atol=5e-6)
    assert np.all(d[:, :, 41:] == 0)
    for b in range(B):
        assert np.all(d[b, int(z["in_len"][b]):] == 0)
    zi = load(golden_dir, "ctc_inf.npz")
    li, _, _ = ops.ctc_loss(torch.from_numpy(zi["logits"]).to(dev), torch.tensor([[4, 4, 4]]), torch.tensor([3]),
                            torch.tensor([3]), False, 1.0, ops.Workspace())
    assert np.isinf(li.cpu().numpy()[0])


def test_ctc_oracle_ragged():
    """Seeded ragged batch incl. S=1, repeats, T_b<T, long targets (2S+1 > 256 states)."""
    import b2t_ops as ops
    rng = np.random.default_rng(3)
    dev = _dev()
    B, T, C, S = 5, 400, 41, 150
    logits = (rng.standard_normal((B, T, C)) * 1.5).astype(np.float32)
    tg = rng.integers(1, C, (B, S)).astype(np.int32)
    tg[0, :6] = [7, 7, 7, 2, 2, 9]
    tl = np.array([150, 1, 60, 33, 120], dtype=np.int32)
    il = np.array([400, 17, 400, 250, 399], dtype=np.int32)
    for b in range(B):
        tg[b, tl[b]:] = 0
    lo, dlo = O.ctc_loss_fwd_bwd(logits, tg, il, tl)
    loss, dl, _ = ops.ctc_loss(torch.from_numpy(logits).to(dev), torch.from_numpy(tg), torch.from_numpy(il),
                               torch.from_numpy(tl), True, 1.0 / B, ops.Workspace())
    np.testing.assert_allclose(loss.cpu().numpy(), lo, rtol=1e-5)
    # long sequences: log-space values reach ~1e3 where one fp32 ulp is 6e-5, so fp32 implementations
    # (this kernel, the oracle, torch CPU) differ from each other at the 1e-4 relative level.  Bound the
    # kernel's error against an fp64 evaluation by a small multiple of the fp32 oracle's own error.
    l64, d64 = O.ctc_loss_fwd_bwd(logits, tg, il, tl, dtype=np.float64)
    got = dl.cpu().numpy()[:, :, :C].astype(np.float64)
    err_gpu = np.abs(got - d64).max()
    err_o32 = np.abs(dlo.astype(np.float64) - d64).max()
    assert err_gpu <= max(4 * err_o32, 5e-6), (err_gpu, err_o32)
    np.testing.assert_allclose(loss.cpu().numpy(), l64, rtol=1e-5)
    np.testing.assert_allclose(got, dlo, atol=5e-6, rtol=2e-3)


def test_train_step_golden(golden_dir):
    """Full fwd + bwd + clip + AdamW for 4 steps vs the reference's step body (rnn_trainer.py:527-558)."""
    from rnn_trainer import TrainStep
    z = load(golden_dir, "train_step.npz")
    dev = _dev()
    m = make_model(z["cfg"], sd_of(z, "sd0::")).train()
    args = dict(lr_max=0.005, lr_min=0.0001, lr_decay_steps=120000, lr_warmup_steps=int(z["warmup"]),
                lr_max_day=0.005, lr_min_day=0.0001, lr_decay_steps_day=120000, lr_warmup_steps_day=int(z["warmup"]),
                beta0=0.9, beta1=0.999, epsilon=0.1, weight_decay=0.001, weight_decay_day=0,
                grad_norm_clip_value=float(z["clip"]), _debug_keep_unclipped=True)
    ts = TrainStep(m, args)
    x = torch.from_numpy(z["x"]).to(dev)
    day = torch.from_numpy(z["day_idx"])
    tgt = torch.from_numpy(z["targets"]); tl = torch.from_numpy(z["tgt_len"]); nts = torch.from_numpy(z["n_time_steps"])
    gold_grads = sd_of(z, "grad0::")
    for it in range(4):
        import b2t_ops as ops
        feats = ops.augment_smooth(x, 2, 100, "same")
        if it == 0:
            np.testing.assert_allclose(feats.cpu().numpy(), z["feats0"], atol=3e-6)
        loss, gnorm = ts.step(feats, day, tgt, nts, tl)
        np.testing.assert_allclose(float(loss), z[f"loss{it}"], rtol=2e-5)
        np.testing.assert_allclose(float(gnorm), z[f"gnorm{it}"], rtol=1e-4)
        if it == 0:
            g = ts.last_unclipped_grads()
            for k, ref in gold_grads.items():
                scale = max(1e-6, float(np.abs(ref).max()))
                np.testing.assert_allclose(g[k], ref, atol=1e-3 * scale, err_msg=k)
        sd = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
        for k, ref in sd_of(z, f"sd{it+1}::").items():
            np.testing.assert_allclose(sd[k], ref, atol=2e-5, err_msg=f"step{it} {k}")


def test_autograd_matches_fused(golden_dir):
    """loss.backward() through GRUDecoder (autograd bridge) == oracle gradients."""
    z = load(golden_dir, "train_step.npz")
    dev = _dev()
    sd0 = sd_of(z, "sd0::")
    m = make_model(z["cfg"], sd0).train()
    feats = torch.from_numpy(z["feats0"]).to(dev)
    logits = m(feats, torch.from_numpy(z["day_idx"]))
    lp = torch.permute(logits.log_softmax(2), [1, 0, 2])
    # an arbitrary differentiable scalar of the logits (torch ops): sum of squares of log-probs
    (lp ** 2).mean().backward()
    L = int(z["cfg"][4])
    lo, _, ctx = O.model_fwd(sd0, z["feats0"], z["day_idx"], L, save=True)
    tl = torch.from_numpy(lo).requires_grad_(True)
    (torch.permute(tl.log_softmax(2), [1, 0, 2]) ** 2).mean().backward()
    dlog = tl.grad.numpy()
    B, Tp, C = dlog.shape
    p = ctx["p"]
    d_out = (dlog.reshape(B * Tp, C) @ p["out_w"]).reshape(B, Tp, -1)
    _, dW_ih, dW_hh, _, _, dh = O.gru_bwd(d_out, ctx["reserve"], p["w_ih"], p["w_hh"], ctx["h_init"])
    for l in range(L):
        for name, ref in ((f"gru.weight_ih_l{l}", dW_ih[l]), (f"gru.weight_hh_l{l}", dW_hh[l])):
            got = dict(m.named_parameters())[name].grad.cpu().numpy()
            np.testing.assert_allclose(got, ref, atol=1e-3 * max(1e-6, np.abs(ref).max()), err_msg=name)
    np.testing.assert_allclose(m.h0.grad.cpu().numpy().reshape(-1), dh.sum((0, 1)), atol=1e-3 * np.abs(dh.sum((0, 1))).max())
    active = set(int(d) for d in z["day_idx"])
    for d in range(int(z["cfg"][2])):
        assert (m.day_weights[d].grad is not None) == (d in active)


def test_greedy_and_edit_golden(golden_dir):
    import b2t_ops as ops
    z = load(golden_dir, "greedy.npz")
    dev = _dev()
    ids, ln, am = ops.greedy_decode(torch.from_numpy(z["logits"]).to(dev), torch.from_numpy(z["lens"]))
    ids, ln = ids.cpu().numpy(), ln.cpu().numpy()
    B = z["logits"].shape[0]
    np.testing.assert_array_equal(am.cpu().numpy(), np.argmax(z["logits"], -1))     # bit-exact argmax indices
    for b in range(B):
        np.testing.assert_array_equal(ids[b, :ln[b]], z[f"trainer_{b}"])
    lab = torch.from_numpy(z["labels"]).to(dev)
    dist = ops.edit_distance(torch.from_numpy(ids).to(dev), torch.from_numpy(ln).to(dev), lab,
                             torch.from_numpy(z["lab_len"]).to(dev)).cpu().numpy()
    np.testing.assert_array_equal(dist, [int(z[f"edit_{b}"]) for b in range(B)])


def test_edit_distance_random():
    import b2t_ops as ops
    rng = np.random.default_rng(0)
    dev = _dev()
    Bn, La, Lb = 40, 150, 130
    a = rng.integers(1, 6, (Bn, La)).astype(np.int32); b = rng.integers(1, 6, (Bn, Lb)).astype(np.int32)
    al = rng.integers(0, La + 1, Bn).astype(np.int32); bl = rng.integers(0, Lb + 1, Bn).astype(np.int32)
    al[0] = 0; bl[1] = 0; al[2] = La; bl[2] = Lb
    got = ops.edit_distance(torch.from_numpy(a).to(dev), torch.from_numpy(al).to(dev), torch.from_numpy(b).to(dev),
                            torch.from_numpy(bl).to(dev)).cpu().numpy()
    ref = [O.edit_distance(a[i, :al[i]], b[i, :bl[i]]) for i in range(Bn)]
    np.testing.assert_array_equal(got, ref)


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("tag", ["h64", "h512", "patch"])
def test_gru_sweep_modes(golden_dir, tag, mode):
    """Step-launch (0) and persistent (1) sweeps both reproduce the reference forward, and the persistent
    hand-off reports no timeout."""
    import b2t_ops as ops
    z = load(golden_dir, f"fwd_{tag}.npz")
    old = ops.GRU_MODE["value"]
    ops.GRU_MODE["value"] = mode
    try:
        m = make_model(z["cfg"], sd_of(z)).eval()
        dev = _dev()
        with torch.no_grad():
            logits, hidden = m(torch.from_numpy(z["x"]).to(dev), torch.from_numpy(z["day_idx"]).to(dev), None, True)
        np.testing.assert_allclose(logits.cpu().numpy(), z["logits"], atol=1e-4)
        np.testing.assert_allclose(hidden.cpu().numpy(), z["hidden"], atol=1e-4)
        if mode >= 1:
            m._ws.check_sync()
    finally:
        ops.GRU_MODE["value"] = old


def N_sync(T):
    import b2t_native as Nn
    return Nn.load().b2t_gru_sync_bytes(T) // 4 + 16


@pytest.mark.parametrize("mode", [0, 1])
def test_train_step_modes_vs_oracle(mode):
    """C2-shaped slice (H=512, L=2, B=40 = 2.5 row groups, T=60): loss + every gradient vs the oracle, both modes,
    run 3 times back-to-back so the persistent hand-off is exercised with warm caches."""
    import b2t_ops as ops
    from rnn_model import GRUDecoder
    from rnn_trainer import TrainStep
    dev = _dev()
    old = ops.GRU_MODE["value"]
    ops.GRU_MODE["value"] = mode
    try:
        torch.manual_seed(3)
        F, H, D, C, L, B, T, S = 64, 512, 4, 41, 2, 40, 60, 7
        model = GRUDecoder(F, H, D, C, 0.0, 0.0, L, 0, 0)
        sd0 = {k: v.detach().numpy().copy() for k, v in model.state_dict().items()}
        model = model.to(dev).train()
        g = torch.Generator().manual_seed(5)
        x = torch.randn(B, T, F, generator=g) * 0.5
        day = torch.randint(0, D, (B,), generator=g)
        tgt = torch.randint(1, C, (B, S), generator=g)
        tl = torch.randint(1, S + 1, (B,), generator=g); nt = torch.randint(30, T + 1, (B,), generator=g)
        for b in range(B):
            tgt[b, tl[b]:] = 0
        lo, _, _, go = O.model_loss_and_grads(sd0, x.numpy(), day.numpy(), tgt.numpy(), nt.numpy(), tl.numpy(), L)
        args = dict(lr_max=0.0, lr_min=0.0, lr_decay_steps=10, lr_warmup_steps=0, lr_max_day=0.0, lr_min_day=0.0,
                    lr_decay_steps_day=10, lr_warmup_steps_day=0, beta0=0.9, beta1=0.999, epsilon=0.1, weight_decay=0.0,
                    weight_decay_day=0, grad_norm_clip_value=0, _debug_keep_unclipped=True)
        args["lr_max"] = 1e-30; args["lr_max_day"] = 1e-30; args["lr_min"] = 1e-30; args["lr_min_day"] = 1e-30
        ts = TrainStep(model, args)
        for rep in range(3):
            loss, gnorm = ts.step(x.to(dev), day, tgt, nt, tl)
            np.testing.assert_allclose(float(loss), float(lo), rtol=2e-5)
            got = ts.last_unclipped_grads()
            assert set(got) == set(go)
            for k, ref in go.items():
                np.testing.assert_allclose(got[k], ref, atol=1e-3 * max(1e-6, float(np.abs(ref).max())), err_msg=f"{k} rep{rep}")
        if mode >= 1:
            model._ws.check_sync()
    finally:
        ops.GRU_MODE["value"] = old


@pytest.mark.parametrize("B,H,T,wide", [(5, 48, 9, 0), (17, 80, 13, 0), (33, 272, 11, 0), (64, 512, 24, 0), (70, 768, 7, 0),
                                         (64, 512, 24, 1), (23, 96, 7, 1), (40, 288, 9, 1)])
def test_persistent_sweep_odd_shapes_match_step_launch(B, H, T, wide):
    """Persistent sweeps (line-wise operand loads + LDS transpose, clamped rows / columns) against the step-launch
    kernels on shapes that are not multiples of the tile sizes: forward out / reserve, backward dG / dh0."""
    import b2t_native as Nn
    import b2t_ops as ops
    lib = Nn.load(); dev = _dev(); p = ops._p
    g = torch.Generator().manual_seed(B * 1000 + H)
    rnd = lambda *s: (torch.randn(*s, generator=g)).to(dev)
    gi, w, b_, h0 = rnd(T, B, 3 * H) * 0.5, rnd(3 * H, H) * (1.0 / H ** 0.5), rnd(3 * H) * 0.1, rnd(B, H) * 0.3
    dY, dhl = rnd(T, B, H) * 0.05, rnd(B, H) * 0.05
    wt = w.t().contiguous()

    def run(mode):
        out = torch.zeros(T + 1, B, H, device=dev); out[0] = h0
        res = torch.zeros(T, B, 4 * H, device=dev)
        sync = torch.zeros(lib.b2t_gru_ws_bytes(T, B, H) // 4 + 16, dtype=torch.int32, device=dev)
        Nn.check(lib.b2t_gru_layer_fwd_f32(p(gi), p(w), p(b_), p(out[0]), p(out[1:]), p(res), None, T, B, H, mode, p(sync),
                                           ops._stream()), "fwd")
        dG = torch.zeros(T, B, 4 * H, device=dev); dh = torch.zeros(B, H, device=dev); sc = torch.empty(B, H, device=dev)
        Nn.check(lib.b2t_gru_layer_bwd_f32(p(dY), p(dhl), p(res), p(out[1:]), p(out[0]), p(wt), p(dG), p(dh), p(sc), T, B, H,
                                           mode, p(sync), ops._stream()), "bwd")
        torch.cuda.synchronize()
        assert int(sync[0]) == 0
        return out, res, dG, dh

    ref = run(0)
    for rep in range(3):
        got = run(1 | (ops.GRU_WIDE if wide else 0))   # wide: 32 hidden units per workgroup, exact fp32
        for a, r, name in zip(got, ref, ("out", "reserve", "dG", "dh0")):
            np.testing.assert_allclose(a.cpu().numpy(), r.cpu().numpy(), atol=3e-6 * max(1.0, float(r.abs().max())), err_msg=name)
    if not wide:
        # ... and the XCD-local hand-off (both parities) equals the device-scope one bit for bit on these shapes too.  (Round 4: with
        # H = 48, B = 5 a cache line held rows of two time steps and the local form, which reads tiles through the vector cache,
        # returned stale values -- shapes with H % 32 != 0 now take the device-scope hand-off.)
        base = run(1)
        for extra in (ops.GRU_LOCAL, ops.GRU_LOCAL | ops.GRU_PARITY):
            got = run(1 | extra)
            for a, r, name in zip(got, base, ("out", "reserve", "dG", "dh0")):
                assert torch.equal(a, r), (name, extra)


@pytest.mark.parametrize("B,H,T", [(64, 512, 24), (40, 288, 9), (23, 96, 7), (17, 32, 5), (64, 256, 130), (33, 384, 11), (64, 480, 6), (5, 64, 12)])
def test_paired_backward_sweep_matches_step_launch(B, H, T):
    """Round 5: the backward sweep with its W_hh^T slice in LDS (B2T_GRU_PAIRED: 512-thread workgroups owning 16 dh columns of two row
    groups, the contraction split over eight waves) against the step-launch kernels -- dG and dh0, every XCD set, repeated calls on
    one sync workspace (the counter sets alternate), odd numbers of row groups (a pair with one live group) and ragged last groups."""
    import b2t_native as Nn
    import b2t_ops as ops
    lib = Nn.load(); dev = _dev(); p = ops._p
    g = torch.Generator().manual_seed(B * 1000 + H)
    rnd = lambda *s: (torch.randn(*s, generator=g)).to(dev)
    gi, w, b_, h0 = rnd(T, B, 3 * H) * 0.5, rnd(3 * H, H) * (1.0 / H ** 0.5), rnd(3 * H) * 0.1, rnd(B, H) * 0.3
    dY, dhl = rnd(T, B, H) * 0.05, rnd(B, H) * 0.05
    wt = w.t().contiguous()
    out = torch.zeros(T + 1, B, H, device=dev); out[0] = h0
    res = torch.zeros(T, B, 4 * H, device=dev)
    sync = torch.zeros(lib.b2t_gru_ws_bytes(T, B, H) // 4 + 16, dtype=torch.int32, device=dev)
    Nn.check(lib.b2t_gru_layer_fwd_f32(p(gi), p(w), p(b_), p(out[0]), p(out[1:]), p(res), None, T, B, H, 0, p(sync), ops._stream()), "fwd")

    def bwd(mode):
        dG = torch.full((T, B, 4 * H), float("nan"), device=dev); dh = torch.full((B, H), float("nan"), device=dev); sc = torch.empty(B, H, device=dev)
        Nn.check(lib.b2t_gru_layer_bwd_f32(p(dY), p(dhl), p(res), p(out[1:]), p(out[0]), p(wt), p(dG), p(dh), p(sc), T, B, H,
                                           mode, p(sync), ops._stream()), "bwd")
        torch.cuda.synchronize()
        assert int(sync[0]) == 0
        return dG, dh

    ref = bwd(0)
    first = None
    for rep in range(2):
        for st in range(4):
            got = bwd(1 | ops.GRU_LOCAL | ops.GRU_PAIRED | (st << ops.GRU_SET_SHIFT))
            for a, r, name in zip(got, ref, ("dG", "dh0")):
                np.testing.assert_allclose(a.cpu().numpy(), r.cpu().numpy(), atol=3e-6 * max(1.0, float(r.abs().max())), err_msg=f"{name} set {st}")
            if first is None:
                first = got
            for a, r in zip(got, first):       # placement changes where the work runs, not the arithmetic
                assert torch.equal(a, r), st


@pytest.mark.parametrize("akc,bkc", [(1, 1), (1, 0), (0, 1), (0, 0)])
@pytest.mark.parametrize("M,N,K", [(128, 128, 128), (200, 41, 37), (33, 300, 129), (512, 1536, 520), (640, 256, 4100),
                                   (300, 2200, 130), (1100, 2304, 200)])     # > 16 tile columns: the grouped tile order (3 and 9 tile rows: one short group, a full + a short one)
def test_gemm_bf16_packed(akc, bkc, M, N, K):
    """b2t_gemm_bf16p_f32 (two passes: pack both operands to dense bf16, then tiles on the packed operands): same contract
    as b2t_gemm_bf16_f32 -- an fp64 product of the bf16-rounded operands up to fp32 summation roundoff -- for all four
    operand layouts, ragged extents, bias / accumulate / Softsign epilogues, split-K slabs, a row-mapped C and a gap in A."""
    import ctypes as C
    import b2t_native as Nn
    import b2t_ops as ops
    lib = Nn.load(); dev = _dev()
    rng = np.random.default_rng(M * 7 + N * 3 + K + 1)
    A = rng.standard_normal((M, K)).astype(np.float32)
    Bm = rng.standard_normal((N, K)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    Aq, Bq = _bf16_round(A).astype(np.float64), _bf16_round(Bm).astype(np.float64)
    ref = Aq @ Bq.T
    Mp, Np, Kp = (M + 3) // 4 * 4, (N + 3) // 4 * 4, (K + 3) // 4 * 4
    if akc:
        Ad = torch.zeros(M, Kp); Ad[:, :K] = torch.from_numpy(A); a_s0 = Kp
    else:
        Ad = torch.zeros(K, Mp); Ad[:, :M] = torch.from_numpy(A.T); a_s0 = Mp
    if bkc:
        Bd = torch.zeros(N, Kp); Bd[:, :K] = torch.from_numpy(Bm); b_s0 = Kp
    else:
        Bd = torch.zeros(K, Np); Bd[:, :N] = torch.from_numpy(Bm.T); b_s0 = Np
    Ad, Bd, bd = Ad.to(dev), Bd.to(dev), torch.from_numpy(bias).to(dev)
    wsb = lib.b2t_gemm_bf16p_ws_bytes(M, N, K)
    ws = torch.empty(wsb // 4 + 64, dtype=torch.float32, device=dev)
    tol = 2e-5 * max(1.0, float(np.abs(ref).max()))

    def run(Cd, **kw):
        d = Nn.GemmDesc()
        d.A, d.B, d.C = Ad.data_ptr(), Bd.data_ptr(), Cd.data_ptr()
        d.M, d.N, d.K, d.Z = M, N, K, 1
        d.a_kcontig, d.b_kcontig, d.a_s0, d.b_s0, d.c_s0 = akc, bkc, a_s0, b_s0, N
        d.splitk = 1
        for k, v in kw.items():
            setattr(d, k, v)
        Nn.check(lib.b2t_gemm_bf16p_f32(C.byref(d), ops._p(ws), wsb, ops._stream()), "b2t_gemm_bf16p_f32")
        return Cd.cpu().numpy()

    got = run(torch.full((M, N), float("nan"), device=dev), bias=bd.data_ptr())
    np.testing.assert_allclose(got, ref + bias, atol=tol)
    if K >= 64:
        exact = A.astype(np.float64) @ Bm.astype(np.float64).T + bias
        assert np.abs(got - exact).max() > 10 * np.abs(got - (ref + bias)).max()       # it is the bf16 product
    np.testing.assert_allclose(run(torch.ones((M, N), device=dev), accumulate=1), ref + 1.0, atol=tol)
    np.testing.assert_allclose(run(torch.zeros((M, N), device=dev), epilogue=1), ref / (1 + np.abs(ref)), atol=tol)
    U = rng.uniform(-0.9, 0.9, size=(M, N)).astype(np.float32); Ud = torch.from_numpy(U).to(dev)
    np.testing.assert_allclose(run(torch.zeros((M, N), device=dev), epilogue=2, ep_aux=Ud.data_ptr()), ref * (1 - np.abs(U)) ** 2, atol=tol)
    # split-K slabs
    sk = 3
    slab = torch.full((sk, M, N), float("nan"), device=dev)
    part = run(slab, splitk=sk, c_ks=M * N)
    np.testing.assert_allclose(part.sum(0), ref, atol=tol)
    if K >= 512:       # 8 slices: the slice-per-XCD block order
        slab8 = torch.full((8, M, N), float("nan"), device=dev)
        np.testing.assert_allclose(run(slab8, splitk=8, c_ks=M * N).sum(0), ref, atol=tol)
    if not akc:
        # by-product of the pack pass over an m-contiguous A: per 64-k slice sums of the fp32 values (a layer's bias gradients)
        nsl = (K + 63) // 64
        sums = torch.full((nsl + 1, M + 2), 7.0, device=dev)
        got = run(torch.empty((M, N), device=dev), splitk=sk, c_ks=M * N, C=slab.data_ptr(), a_sum=sums.data_ptr(), a_sum_ks=M + 2)
        sg = sums.cpu().numpy()
        assert np.all(sg[nsl:] == 7.0) and np.all(sg[:, M:] == 7.0)
        for s_ in range(nsl):
            np.testing.assert_allclose(sg[s_, :M], A[:, s_ * 64:(s_ + 1) * 64].astype(np.float64).sum(1), atol=1e-4)
    # row-mapped C (batch-first logits: row r = t * Bb + b -> C[b][t]) and a gap in A's contiguous index
    if M % 4 == 0:
        Bb = 4; Tt = M // Bb
        Cm = torch.full((Bb, Tt, N), float("nan"), device=dev)
        out = run(Cm, c_div=Bb, c_s1=N, c_s0=Tt * N)
        np.testing.assert_allclose(out.transpose(1, 0, 2).reshape(M, N), ref, atol=tol)
    if akc and K % 128 == 0 or (not akc and M % 256 == 0):
        # A' = A with `gap` extra elements spliced into the contiguous index at brk
        gap = 8
        if akc:
            brk = K // 2
            Ag = torch.full((M, Kp + gap), 7.0); Ag[:, :brk] = torch.from_numpy(A[:, :brk]); Ag[:, brk + gap:K + gap] = torch.from_numpy(A[:, brk:])
            kw = dict(a_s0=Kp + gap)
        else:
            brk = M // 2
            Ag = torch.full((K, Mp + gap), 7.0); Ag[:, :brk] = torch.from_numpy(A.T[:, :brk]); Ag[:, brk + gap:M + gap] = torch.from_numpy(A.T[:, brk:])
            kw = dict(a_s0=Mp + gap)
        Agd = Ag.to(dev)
        d_keep = Ad
        Ad = Agd
        try:
            got = run(torch.full((M, N), float("nan"), device=dev), a_brk=brk, a_gap=gap, **kw)
        finally:
            Ad = d_keep
        np.testing.assert_allclose(got, ref, atol=tol)


@pytest.mark.parametrize("form", ["day_forward", "day_wgrad"])
@pytest.mark.parametrize("Z,T,F", [(5, 83, 128), (64, 200, 256), (7, 130, 72)])
def test_gemm_bf16_packed_z_batched(form, Z, T, F):
    """b2t_gemm_bf16p_f32 with a Z-batched descriptor (round 5): the day layer's per-sentence products -- forward
    U[b] = softsign(x[b] W[day[b]] + c[day[b]]) (A k-contiguous, B = the day's weight selected by b_zmap and stored k-major,
    bias by bias_sz, Softsign epilogue; rnn_model.py:95-99) and backward dW[b] = x[b]^T dpre[b] (both operands m-contiguous,
    K = T, per-sentence slabs) -- every matrix of the batch packed, one launch over (tiles, Z).  Against the fp64 product of the
    bf16-rounded operands; the Softsign-backward epilogue (ep_aux per z) and accumulate too."""
    import ctypes as C
    import b2t_native as Nn
    import b2t_ops as ops
    lib = Nn.load(); dev = _dev()
    rng = np.random.default_rng(Z * 100 + T + F)
    D = 3
    x = rng.standard_normal((Z, T, F)).astype(np.float32)
    W = (rng.standard_normal((D, F, F)) / np.sqrt(F)).astype(np.float32)       # [day][k][n]
    bias = rng.standard_normal((D, F)).astype(np.float32)
    day = rng.integers(0, D, Z).astype(np.int32)
    dpre = rng.standard_normal((Z, T, F)).astype(np.float32)
    xq, Wq, dq = _bf16_round(x).astype(np.float64), _bf16_round(W).astype(np.float64), _bf16_round(dpre).astype(np.float64)
    xd, Wd, bd, dayd, dd = (torch.from_numpy(a).to(dev) for a in (x, W, bias, day, dpre))
    d = Nn.GemmDesc()
    if form == "day_forward":
        M, N, K = T, F, F
        ref = np.einsum("ztk,zkn->ztn", xq, Wq[day]) + bias[day][:, None, :]
        out = torch.full((Z, T, F), float("nan"), device=dev)
        d.A, d.B, d.C = xd.data_ptr(), Wd.data_ptr(), out.data_ptr()
        d.a_kcontig, d.a_s0, d.a_sz = 1, F, T * F
        d.b_kcontig, d.b_s0, d.b_sz, d.b_zmap = 0, F, F * F, dayd.data_ptr()
        d.c_s0, d.c_sz, d.bias, d.bias_sz = F, T * F, bd.data_ptr(), F
    else:
        M, N, K = F, F, T
        ref = np.einsum("ztm,ztn->zmn", xq, dq)
        out = torch.full((Z, F, F), float("nan"), device=dev)
        d.A, d.B, d.C = xd.data_ptr(), dd.data_ptr(), out.data_ptr()
        d.a_kcontig, d.a_s0, d.a_sz = 0, F, T * F
        d.b_kcontig, d.b_s0, d.b_sz = 0, F, T * F
        d.c_s0, d.c_sz = F, F * F
    d.M, d.N, d.K, d.Z, d.splitk = M, N, K, Z, 1
    wsb = lib.b2t_gemm_bf16p_ws_bytes_z(M, N, K, Z)
    assert wsb >= Z * (lib.b2t_gemm_bf16p_ws_bytes(M, N, K) - 512)
    ws = torch.empty(wsb // 4 + 64, dtype=torch.float32, device=dev)
    tol = 2e-5 * max(1.0, float(np.abs(ref).max()))

    def run(**kw):
        for k, v in kw.items():
            setattr(d, k, v)
        Nn.check(lib.b2t_gemm_bf16p_f32(C.byref(d), ops._p(ws), wsb, ops._stream()), "b2t_gemm_bf16p_f32")
        return out.cpu().numpy()

    np.testing.assert_allclose(run(), ref, atol=tol)
    one = torch.full_like(out, float("nan"))                    # the one-pass kernel: same contract
    d.C = one.data_ptr()
    Nn.check(lib.b2t_gemm_bf16_f32(C.byref(d), ops._stream()), "b2t_gemm_bf16_f32")
    np.testing.assert_allclose(one.cpu().numpy(), ref, atol=tol)
    d.C = out.data_ptr()
    np.testing.assert_allclose(run(epilogue=1), ref / (1 + np.abs(ref)), atol=tol)
    U = rng.uniform(-0.9, 0.9, size=ref.shape).astype(np.float32); Ud = torch.from_numpy(U).to(dev)
    np.testing.assert_allclose(run(epilogue=2, ep_aux=Ud.data_ptr()), ref * (1 - np.abs(U)) ** 2, atol=tol)
    out.fill_(1.0)
    np.testing.assert_allclose(run(epilogue=0, ep_aux=None, accumulate=1), ref + 1.0, atol=tol)
    # too small a workspace is refused
    assert lib.b2t_gemm_bf16p_f32(C.byref(d), ops._p(ws), lib.b2t_gemm_bf16p_ws_bytes(M, N, K), ops._stream()) != 0


@pytest.mark.parametrize("M,N,K", [(2200, 2304, 200), (520, 7168, 130), (2304, 7168, 192)])
def test_gemm_bf16_packed_256_tiles(M, N, K, monkeypatch):
    """The 256 x 256 kernel of b2t_gemm_bf16p_f32 (gemm_bf16p_kernel256: chosen where its tiles fill the chip evenly;
    B2T_GEMM_256 = 2 forces it from 64 tiles on, 0 disables it; read per call): every output element sums the same products in
    the same order as in the 128 x 128 kernel, so the two are BIT-identical -- ragged extents (operand rows clamped where the
    packed matrix, padded to 128 rows, ends inside a 256-row tile), bias / accumulate / Softsign epilogues, split-K slabs incl.
    the slice-per-XCD order, a row-mapped C -- and both are the bf16 product."""
    import ctypes as C
    import b2t_native as Nn
    import b2t_ops as ops
    lib = Nn.load(); dev = _dev()
    g = torch.Generator().manual_seed(M + N + K)
    Ks = (K + 3) // 4 * 4 + 4                     # row stride (a multiple of 4 elements, not K)
    A = torch.randn(M, Ks, generator=g); Bm = torch.randn(N, Ks, generator=g); bias = torch.randn(N, generator=g)
    ref = _bf16_round(A[:, :K].numpy()).astype(np.float64) @ _bf16_round(Bm[:, :K].numpy()).astype(np.float64).T
    Ad, Bd, bd = A.to(dev), Bm.to(dev), bias.to(dev)
    U = (torch.rand(M, N, generator=g) * 1.8 - 0.9).to(dev)
    wsb = lib.b2t_gemm_bf16p_ws_bytes(M, N, K)
    ws = torch.empty(wsb // 4 + 64, dtype=torch.float32, device=dev)

    def run(Cd, **kw):
        d = Nn.GemmDesc()
        d.A, d.B, d.C = Ad.data_ptr(), Bd.data_ptr(), Cd.data_ptr()
        d.M, d.N, d.K, d.Z = M, N, K, 1
        d.a_kcontig, d.b_kcontig, d.a_s0, d.b_s0, d.c_s0 = 1, 1, Ks, Ks, N
        d.splitk = 1
        for k, v in kw.items():
            setattr(d, k, v)
        Nn.check(lib.b2t_gemm_bf16p_f32(C.byref(d), ops._p(ws), wsb, ops._stream()), "b2t_gemm_bf16p_f32")
        return Cd.clone()

    def all_forms():
        out = [run(torch.full((M, N), float("nan"), device=dev), bias=bd.data_ptr()), run(torch.ones((M, N), device=dev), accumulate=1),
               run(torch.zeros((M, N), device=dev), epilogue=1), run(torch.zeros((M, N), device=dev), epilogue=2, ep_aux=U.data_ptr()),
               run(torch.full((3, M, N), float("nan"), device=dev), splitk=3, c_ks=M * N)]
        out.append(run(torch.full((8, M, N), float("nan"), device=dev), splitk=8, c_ks=M * N))
        if M % 4 == 0:
            out.append(run(torch.full((4, M // 4, N), float("nan"), device=dev), c_div=4, c_s1=N, c_s0=(M // 4) * N))
        return out

    monkeypatch.setenv("B2T_GEMM_256", "0"); small = all_forms()
    monkeypatch.setenv("B2T_GEMM_256", "2"); big = all_forms()
    monkeypatch.delenv("B2T_GEMM_256"); auto = all_forms()
    for i, (a, b, c) in enumerate(zip(small, big, auto)):
        assert torch.equal(a, b) and torch.equal(a, c), i
    tol = 2e-5 * max(1.0, float(np.abs(ref).max()))
    np.testing.assert_allclose(big[0].cpu().numpy(), ref + bias.numpy(), atol=tol)
    np.testing.assert_allclose(big[4].cpu().numpy().sum(0), ref, atol=tol)
    np.testing.assert_allclose(big[5].cpu().numpy().sum(0), ref, atol=tol)


@pytest.mark.parametrize("M,N,K,sk,brk", [(256, 128, 1000, 1, 0), (384, 200, 4100, 5, 0), (130, 64, 700, 2, 0), (1536, 512, 6000, 21, 1024),
                                          (100, 300, 50, 1, 0), (768, 130, 40000, 30, 512), (256, 300, 9000, 16, 0), (384, 128, 5000, 8, 0), (1536, 512, 8000, 24, 1024)])
def test_gemm_a_column_sums(M, N, K, sk, brk):
    """b2t_gemm_f32, m-contiguous A (the weight-gradient form dW = dG^T X): `a_sum` receives, per K slice, the sums over k
    of A[k][m] -- the bias gradients of a GRU layer (torch autograd of nn.GRU's b_ih / b_hh: column sums of the gate
    gradients) as a by-product of the GEMM that reads dG anyway.  Checked against float64 sums, with split-K, with the
    gap in A's m index (dGi = dG[:, 0:2H] ++ dG[:, 3H:4H]), ragged M and N, and nothing written outside [slices][M].
    Slice counts that are multiples of 8 take the slice-per-XCD block order."""
    import b2t_ops as ops
    dev = _dev()
    rng = np.random.default_rng(M + N + K)
    gap = 64 if brk else 0
    At = (rng.standard_normal((K, (M + gap + 3) // 4 * 4)) + 0.25).astype(np.float32)      # [K][M (+gap), padded to 4]: m contiguous
    Bt = np.zeros((K, (N + 3) // 4 * 4), np.float32); Bt[:, :N] = rng.standard_normal((K, N))
    cols = np.r_[0:brk, brk + gap:M + gap] if brk else np.arange(M)
    Al = At[:, cols].astype(np.float64)
    ref = Al.T @ Bt[:, :N].astype(np.float64)
    tA, tB = torch.from_numpy(At).to(dev), torch.from_numpy(Bt).to(dev)
    ld = At.shape[1]
    ns = max(1, sk)
    sums = torch.full((ns + 1, M + 3), 7.0, device=dev)
    tC = torch.empty((M, N), device=dev)
    ws = ops.Workspace() if hasattr(ops, "Workspace") else None
    kw = dict(M=M, N_=N, K=K, a_kc=0, b_kc=0, a_s0=ld, b_s0=Bt.shape[1], c_s0=N, a_brk=brk, a_gap=gap, a_sum=sums, a_sum_ks=M + 3)
    if sk > 1:
        ops.gemm(tA, tB, tC, splitk=sk, ws=ws, **kw)
    else:
        ops.gemm(tA, tB, tC, **kw)
    got = sums.cpu().numpy()
    tol = 3e-6 * np.sqrt(K) * max(1.0, float(np.abs(ref).max()))
    np.testing.assert_allclose(tC.cpu().numpy(), ref, atol=tol)
    np.testing.assert_allclose(got[:ns, :M].astype(np.float64).sum(0), Al.sum(0), atol=2e-6 * K ** 0.5 * (1 + np.abs(Al).sum(0).max() / K ** 0.5))
    assert np.all(got[ns:] == 7.0) and np.all(got[:, M:] == 7.0)
    # per-slice sums: slice s covers k in [s*kc, (s+1)*kc), kc = ceil(K / ns) rounded up to 16
    kc = -(-(-(-K // ns)) // 16) * 16
    for s_ in range(ns):
        np.testing.assert_allclose(got[s_, :M], Al[s_ * kc:(s_ + 1) * kc].sum(0), atol=1e-3 + 1e-5 * kc)


@pytest.mark.parametrize("M,N,K,sk,akc", [(1536, 512, 6000, 16, 0), (384, 200, 4100, 5, 0), (130, 64, 700, 2, 1), (256, 256, 9000, 24, 1), (2304, 768, 3000, 7, 0)])
def test_gemm_splitk_in_kernel_reduction(M, N, K, sk, akc):
    """b2t_gemm_f32 split-K with `ks_counters`: the last slice workgroup of every tile sums the slabs inside the GEMM.
    Bit-identical to slabs + b2t_slab_reduce_f32 (same slice order), with and without accumulation, counters left at zero,
    repeated calls on the same counters."""
    import ctypes as C
    import b2t_native as Nn
    import b2t_ops as ops
    lib = Nn.load(); dev = _dev()
    rng = np.random.default_rng(M + 3 * N + K)
    A = rng.standard_normal((M, K)).astype(np.float32); Bm = rng.standard_normal((N, K)).astype(np.float32)
    Mp = (M + 3) // 4 * 4
    if akc:
        Ad = torch.from_numpy(A).to(dev); a_s0 = K
    else:
        At = np.zeros((K, Mp), np.float32); At[:, :M] = A.T; Ad = torch.from_numpy(At).to(dev); a_s0 = Mp
    Bd = torch.from_numpy(np.ascontiguousarray(Bm.T)).to(dev)           # [K][N]: n contiguous
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    cnt = torch.zeros(tiles + 8, dtype=torch.int32, device=dev)

    def run(out, fused, acc):
        slab = torch.full((sk, M, N), float("nan"), device=dev)
        d = Nn.GemmDesc()
        d.A, d.B, d.C = Ad.data_ptr(), Bd.data_ptr(), slab.data_ptr()
        d.M, d.N, d.K, d.Z = M, N, K, 1
        d.a_kcontig, d.b_kcontig, d.a_s0, d.b_s0, d.c_s0 = akc, 0, a_s0, N, N
        d.splitk, d.c_ks = sk, M * N
        if fused:
            d.ks_counters, d.ks_out, d.ks_accumulate = cnt.data_ptr(), out.data_ptr(), acc
        Nn.check(lib.b2t_gemm_f32(C.byref(d), ops._stream()), "b2t_gemm_f32")
        if not fused:
            Nn.check(lib.b2t_slab_reduce_f32(ops._p(slab), sk, M * N, ops._p(out), acc, ops._stream()), "b2t_slab_reduce_f32")
        return out.cpu().numpy()

    base = torch.from_numpy(rng.standard_normal((M, N)).astype(np.float32)).to(dev)
    for acc in (0, 1, 1, 0):
        want = run(base.clone(), False, acc)
        got = run(base.clone(), True, acc)
        assert np.array_equal(got, want)
        assert int(cnt.abs().max().item()) == 0
    ref = A.astype(np.float64) @ Bm.astype(np.float64).T
    np.testing.assert_allclose(run(base.clone(), True, 0), ref, atol=3e-6 * np.sqrt(K) * max(1.0, float(np.abs(ref).max())))


@pytest.mark.parametrize("M,N,K", [(1, 256, 16), (16, 300, 768), (17, 2304, 768), (32, 2304, 7168), (50, 512, 1040), (64, 257, 64), (32, 41, 768), (3, 7, 32)])
def test_gemm_skinny_rows(M, N, K):
    """b2t_gemm_f32 with <= 64 rows against a wide weight matrix (one streamed frame: evaluate_model_helpers.py:87-115)
    takes the weight-streaming kernel (gemm_skinny_kernel): same product, bias, Softsign, accumulate, a row-mapped A and C,
    nothing written outside the extent."""
    import b2t_ops as ops
    dev = _dev()
    rng = np.random.default_rng(M * 31 + N + K)
    A = rng.standard_normal((M, K)).astype(np.float32)
    Bm = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    ref = A.astype(np.float64) @ Bm.astype(np.float64).T
    tA, tB, tb = torch.from_numpy(A).to(dev), torch.from_numpy(Bm).to(dev), torch.from_numpy(bias).to(dev)
    tol = 3e-6 * np.sqrt(K) * max(1.0, float(np.abs(ref).max()))
    Np = N + 5
    tC = torch.full((M + 1, Np), 7.0, device=dev)
    ops.gemm(tA, tB, tC, M=M, N_=N, K=K, a_s0=K, b_s0=K, c_s0=Np, bias=tb)
    got = tC.cpu().numpy()
    np.testing.assert_allclose(got[:M, :N], ref + bias, atol=tol)
    assert np.all(got[M:, :] == 7.0) and np.all(got[:, N:] == 7.0)          # no out-of-bounds writes
    tC2 = torch.ones((M, N), device=dev)
    ops.gemm(tA, tB, tC2, M=M, N_=N, K=K, a_s0=K, b_s0=K, c_s0=N, accumulate=1)
    np.testing.assert_allclose(tC2.cpu().numpy(), ref + 1.0, atol=tol)
    tC3 = torch.zeros((M, N), device=dev)
    ops.gemm(tA, tB, tC3, M=M, N_=N, K=K, a_s0=K, b_s0=K, c_s0=N, epilogue=1)
    np.testing.assert_allclose(tC3.cpu().numpy(), ref / (1 + np.abs(ref)), atol=tol)
    if M % 4 == 0:
        # rows r = t * Bb + b read from A stored [Bb][Tt][K] (the streamed frame's (t, b) row map) and written batch-first
        Bb = 4; Tt = M // Bb
        A3 = np.ascontiguousarray(A.reshape(Tt, Bb, K).transpose(1, 0, 2))
        tA3 = torch.from_numpy(A3).to(dev)
        tC4 = torch.full((Bb, Tt, N), float("nan"), device=dev)
        ops.gemm(tA3, tB, tC4, M=M, N_=N, K=K, a_div=Bb, a_s1=K, a_s0=Tt * K, b_s0=K, c_div=Bb, c_s1=N, c_s0=Tt * N)
        np.testing.assert_allclose(tC4.cpu().numpy().transpose(1, 0, 2).reshape(M, N), ref, atol=tol)


def test_gemm_skinny_batched_day_form():
    """The day layer's per-sentence product on a short frame window (rnn_model.py:95-99): Z sentences x M <= 64 frames, weights
    and bias picked per sentence through b_zmap, W stored [k][n] (b_kcontig = 0), Softsign epilogue -- the batched form of the
    weight-streaming kernel."""
    import b2t_ops as ops
    dev = _dev()
    rng = np.random.default_rng(3)
    Z, M, N, K, D = 5, 14, 48, 64, 3
    x = rng.standard_normal((Z, M, K)).astype(np.float32)
    W = (rng.standard_normal((D, K, N)) * 0.2).astype(np.float32)
    b = rng.standard_normal((D, N)).astype(np.float32)
    zmap = np.array([2, 0, 1, 2, 0], np.int32)
    pre = np.einsum("zmk,zkn->zmn", x.astype(np.float64), W[zmap].astype(np.float64)) + b[zmap][:, None, :]
    ref = pre / (1 + np.abs(pre))
    tx, tW, tb = torch.from_numpy(x).to(dev), torch.from_numpy(W).to(dev), torch.from_numpy(b).to(dev)
    out = torch.full((Z, M, N), float("nan"), device=dev)
    ops.gemm(tx, tW, out, M=M, N_=N, K=K, Z=Z, a_kc=1, a_s0=K, a_sz=M * K, b_kc=0, b_s0=N, b_sz=K * N, c_s0=N, c_sz=M * N,
             bias=tb, bias_sz=N, b_zmap=torch.from_numpy(zmap).to(dev), epilogue=1)
    np.testing.assert_allclose(out.cpu().numpy(), ref, atol=2e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("B,F,H,L,patch,stride,frames,per_call", [(32, 512, 768, 5, 14, 4, 3, 1), (5, 64, 64, 2, 0, 0, 4, 1),
                                                                    (3, 32, 96, 5, 14, 4, 3, 2), (40, 64, 128, 3, 4, 2, 2, 3)])
def test_stream_forward_equals_executor(B, F, H, L, patch, stride, frames, per_call):
    """The opt-in fused streaming frame (csrc/stream.hip, b2t_stream_forward_f32: day layer, patch, L GRU steps and the head in
    one persistent launch with grid barriers) returns the executor path's logits and carried states (rnn_model.py:88-134 with
    `states`), frame after frame: 1 / 2 / 4 batch tiles, with and without patching, several output frames per call."""
    import b2t_ops as ops
    from rnn_model import GRUDecoder
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    model = GRUDecoder(F, H, 4, 41, 0.0, 0.0, L, patch, stride).to(dev).eval()
    day = (torch.arange(B, dtype=torch.int32, device=dev) % 4)
    T_all = (patch if patch else 1) + (stride if patch else 1) * (frames * per_call - 1)
    x_all = torch.randn(B, T_all, F, device=dev) * 0.5
    outs = {}
    was = ops.STREAM["fused"]
    try:
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
            outs[fused] = (torch.cat(got, 1), states)
    finally:
        ops.STREAM["fused"] = was
    assert not torch.isnan(outs[True][0]).any()
    assert float((outs[True][0] - outs[False][0]).abs().max()) < 5e-6
    assert float((outs[True][1] - outs[False][1]).abs().max()) < 5e-6


@pytest.mark.gpu
@pytest.mark.parametrize("S", [1, 7, 60, 64, 100, 128, 150, 255, 300])
def test_ctc_wave_recursion_equals_thread_per_state(S, monkeypatch):
    """The alpha / beta recursions inside one wave (2 / 4 / 8 states per lane, neighbours through whole-wave DPP shifts; up to 512
    extended states) against the thread-per-state form with LDS rows and barriers (B2T_CTC_WAVE=0, also what longer label sequences
    still use): the same operations in the same order, so losses and gradients must be IDENTICAL, ragged lengths, repeats and an
    infeasible sentence included."""
    import b2t_ops as ops
    rng = np.random.default_rng(100 + S)
    dev = _dev()
    B, T, C = 6, 2 * S + 40, 41
    logits = torch.from_numpy((rng.standard_normal((B, T, C)) * 1.5).astype(np.float32)).to(dev)
    tg = rng.integers(1, C, (B, S)).astype(np.int32)
    tl = np.array([S, 1, max(1, S // 2), max(1, S - 1), S, max(1, S // 3)], dtype=np.int32)
    il = np.array([T, min(T, 17), T, T - 3, S, T], dtype=np.int32)       # sentence 4: T_b = S frames: infeasible with repeats
    if S >= 3:
        tg[0, :3] = [7, 7, 7]
        tg[4, :2] = [5, 5]
    for b in range(B):
        tg[b, tl[b]:] = 0
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("B2T_CTC_WAVE", mode)
        loss, dl, _ = ops.ctc_loss(logits, torch.from_numpy(tg), torch.from_numpy(il), torch.from_numpy(tl), True, 1.0 / B, ops.Workspace())
        out[mode] = (loss.cpu().numpy().copy(), dl.cpu().numpy().copy())
    assert np.array_equal(out["0"][0], out["1"][0], equal_nan=True), (out["0"][0], out["1"][0])
    fin = np.isfinite(out["0"][0])
    assert fin.sum() >= 4
    assert np.array_equal(out["0"][1][fin], out["1"][1][fin])
    lo, _ = O.ctc_loss_fwd_bwd(logits.cpu().numpy(), tg, il, tl)
    np.testing.assert_allclose(out["1"][0][fin], lo[fin], rtol=2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("B,F,H,L,patch,stride", [(32, 512, 768, 5, 14, 4), (3, 32, 96, 2, 0, 0)])
def test_streaming_calls_replayed_as_graphs(B, F, H, L, patch, stride):
    """From its third call of a shape on, an eval-mode streaming call (forward with carried states, <= 8 output frames) is
    replayed as one hipGraph around the same executor call (rnn_model._graph_forward): bit-identical to the eager calls frame
    after frame, results stay valid across later calls (they are copies), parameter updates between calls are seen."""
    import b2t_ops as ops
    from rnn_model import GRUDecoder
    dev = torch.device("cuda:0")
    torch.manual_seed(4)
    model = GRUDecoder(F, H, 4, 41, 0.0, 0.0, L, patch, stride).to(dev).eval()
    day = (torch.arange(B, dtype=torch.int32, device=dev) % 4)
    frames = 7
    T_all = (patch if patch else 1) + (stride if patch else 1) * (frames - 1)
    x_all = torch.randn(B, T_all, F, device=dev) * 0.5

    def run(use_graph, bump_at=None):
        was = ops.STREAM["graph"]
        ops.STREAM["graph"] = use_graph
        model._graphs.clear()
        bias0 = model.out.bias.data.clone()
        try:
            states, outs, sts = None, [], []
            with torch.no_grad():
                for f in range(frames):
                    if f == bump_at:
                        model.out.bias.data.add_(0.5)          # a parameter update between two streaming calls
                    xf = (x_all[:, f * stride: f * stride + patch] if patch else x_all[:, f:f + 1]).contiguous()
                    lg, states = model(xf, day, states, True)
                    outs.append(lg); sts.append(states)        # kept WITHOUT cloning: later calls must not overwrite them
            torch.cuda.synchronize()
            return [o.clone() for o in outs], [s.clone() for s in sts]
        finally:
            ops.STREAM["graph"] = was
            model.out.bias.data.copy_(bias0)                  # exactly (b + 0.5 - 0.5 need not be b)

    eager_l, eager_s = run(False)
    graph_l, graph_s = run(True)
    assert any(e.get("graph") is not None for e in model._graphs.values()), "no graph was captured"
    for f in range(frames):
        assert torch.equal(eager_l[f], graph_l[f]) and torch.equal(eager_s[f], graph_s[f]), f
    bump_e, _ = run(False, bump_at=5)
    bump_g, _ = run(True, bump_at=5)
    for f in range(frames):
        assert torch.equal(bump_e[f], bump_g[f]), f
    assert float((bump_g[5] - graph_l[5]).abs().max()) > 0.4     # the update did reach the replayed graph


@pytest.mark.gpu
def test_copy_segments_and_indirect_table():
    """b2t_copy_segments_b32 (up to four device copies in one launch) and b2t_copy_indirect_b32 (the same with the segment list read
    from a pinned table when the kernel runs: rewritten between two launches, the second launch copies the new segments)."""
    import ctypes as C
    import b2t_native as N
    import b2t_ops as ops
    dev = _dev()
    lib = N.load()
    g = torch.Generator().manual_seed(1)
    src = [torch.randn(n, generator=g).to(dev) for n in (1, 1000, 70001)] + [torch.arange(257, dtype=torch.int32, device=dev)]
    dst = [torch.zeros_like(t) for t in src]
    ops.copy_segments(list(zip(src, dst)))
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(src, dst))
    with pytest.raises(RuntimeError):
        ops.copy_segments([(src[0], dst[1])])                       # sizes differ
    tab = torch.zeros(16, dtype=torch.int64).pin_memory()
    t = tab.numpy()
    out = [torch.zeros_like(x) for x in src[:2]]
    t[0] = 2
    for k in range(2):
        t[1 + k], t[5 + k], t[9 + k] = src[k].data_ptr(), out[k].data_ptr(), src[k].numel()
    N.check(lib.b2t_copy_indirect_b32(C.c_void_p(tab.data_ptr()), 64, ops._stream()), "b2t_copy_indirect_b32")
    torch.cuda.synchronize()
    assert torch.equal(out[0], src[0]) and torch.equal(out[1], src[1])
    out2 = torch.zeros_like(src[2])
    t[0] = 1; t[1], t[5], t[9] = src[2].data_ptr(), out2.data_ptr(), src[2].numel()
    N.check(lib.b2t_copy_indirect_b32(C.c_void_p(tab.data_ptr()), 64, ops._stream()), "b2t_copy_indirect_b32")
    torch.cuda.synchronize()
    assert torch.equal(out2, src[2])


@pytest.mark.parametrize("ps,st", [(14, 4), (0, 0), (5, 3), (1, 7)])
def test_adjusted_lens_kernel_equals_the_reference_expression(ps, st):
    """b2t_adjusted_lens_i32 against rnn_trainer.py:532's expression evaluated by torch (fp32 division, truncation), every length
    from patch_size to 4000, int32 and int64 inputs."""
    import b2t_native as Nn
    import b2t_ops as ops
    lib = Nn.load(); dev = _dev()
    n = torch.arange(max(ps, 1), 4000, dtype=torch.int64)
    ref = ((n - ps).to(torch.float32) / st + 1).to(torch.int32) if ps > 0 else n.to(torch.int32)
    for dt in (torch.int32, torch.int64):
        nd = n.to(dt).to(dev)
        out = torch.empty(n.numel(), dtype=torch.int32, device=dev)
        Nn.check(lib.b2t_adjusted_lens_i32(ops._p(nd), int(dt == torch.int64), n.numel(), ps, st, ops._p(out), ops._stream()), "adj")
        assert torch.equal(out.cpu(), ref)


@pytest.mark.parametrize("B,T,F,ps,st,p_drop", [(3, 61, 48, 14, 4, 0.2), (2, 30, 16, 4, 2, 0.0), (5, 100, 512, 14, 4, 0.2), (1, 17, 32, 6, 3, 0.5)])
def test_patch_fold_day_bwd_equals_the_three_kernels(B, T, F, ps, st, p_drop):
    import ctypes as C
    """b2t_patch_fold_day_bwd_f32 (fold + input-dropout backward + Softsign backward, one pass) against b2t_patch_fold_f32 ->
    b2t_dropout_f32 -> b2t_softsign_bwd_f32: every element goes through the same operations in the same order -- bit-identical."""
    import b2t_native as Nn
    import b2t_ops as ops
    lib = Nn.load(); dev = _dev(); p = ops._p
    Tp = (T - ps) // st + 1
    g = torch.Generator().manual_seed(B * 100 + T)
    dv = torch.randn(B, Tp, ps * F, generator=g).to(dev)
    u = (torch.rand(B, T, F, generator=g) * 1.8 - 0.9).to(dev)
    seed = 123456789
    a = torch.empty(B, T, F, device=dev)
    Nn.check(lib.b2t_patch_fold_f32(p(dv), p(a), B, T, F, Tp, ps, st, ops._stream()), "fold")
    if p_drop > 0:
        Nn.check(lib.b2t_dropout_f32(p(a), p(a), B * T * F, C.c_float(p_drop), seed, 0, ops._stream()), "dropout")
    Nn.check(lib.b2t_softsign_bwd_f32(p(u), p(a), B * T * F, ops._stream()), "softsign")
    b = torch.empty(B, T, F, device=dev)
    Nn.check(lib.b2t_patch_fold_day_bwd_f32(p(dv), p(u), p(b), B, T, F, Tp, ps, st, C.c_float(p_drop), seed, ops._stream()), "fused")
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    if p_drop > 0:
        assert float((b == 0).float().mean()) > 0.5 * p_drop
