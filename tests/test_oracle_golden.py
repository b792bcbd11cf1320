"""The oracle (oracle/b2t_oracle.py) pinned against golden vectors captured from the reference
(tests/golden/make_golden.py) and against the reference's one known-answer test
(language_model/runtime/core/decoder/ctc_prefix_beam_search_test.cc:18-59).  CPU only."""
import math
import os

import numpy as np
import pytest

from oracle import b2t_oracle as O


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def sd_of(z, prefix="sd::"):
    return {k[len(prefix):]: z[k] for k in z.files if k.startswith(prefix)}


def test_gauss_kernel_taps(golden_dir):
    z = load(golden_dir, "smooth.npz")
    k = O.gauss_kernel(2, 100)
    assert k.shape == (9,)
    np.testing.assert_allclose(k, z["taps"], rtol=0, atol=1e-8)
    # values printed in SURVEY §0 fact 7
    np.testing.assert_allclose(k[:5], [0.02763055, 0.06628225, 0.12383153, 0.18017381, 0.20416369], atol=1e-7)


def test_gauss_smooth_same_valid(golden_dir):
    z = load(golden_dir, "smooth.npz")
    np.testing.assert_allclose(O.gauss_smooth(z["x"], 2, 100, "same"), z["same"], atol=2e-6)
    np.testing.assert_allclose(O.gauss_smooth(z["x"], 2, 100, "valid"), z["valid"], atol=2e-6)
    np.testing.assert_allclose(O.gauss_smooth(z["x2"], 1, 50, "same"), z["same_std1"], atol=2e-6)


@pytest.mark.parametrize("tag", ["h64", "h512", "patch", "f512"])
def test_model_forward(golden_dir, tag):
    z = load(golden_dir, f"fwd_{tag}.npz")
    F, H, D, C, L, ps, st = [int(v) for v in z["cfg"]]
    sd = sd_of(z)
    logits, hidden = O.model_fwd(sd, z["x"], z["day_idx"], L, ps, st)
    np.testing.assert_allclose(logits, z["logits"], atol=2e-5, rtol=0)
    np.testing.assert_allclose(hidden, z["hidden"], atol=2e-5, rtol=0)
    if "stream_split" in z.files:
        t1 = int(z["stream_split"])
        l1, s1 = O.model_fwd(sd, z["x"][:, :t1], z["day_idx"], L, ps, st)
        l2, s2 = O.model_fwd(sd, z["x"][:, t1:], z["day_idx"], L, ps, st, states=s1)
        np.testing.assert_allclose(np.concatenate([l1, l2], 1), z["stream_logits"], atol=2e-5)
        np.testing.assert_allclose(s2, z["stream_state"], atol=2e-5)
        # streaming == whole-sequence
        np.testing.assert_allclose(np.concatenate([l1, l2], 1), logits, atol=2e-5)


def test_ctc(golden_dir):
    z = load(golden_dir, "ctc.npz")
    loss, dlogits = O.ctc_loss_fwd_bwd(z["logits"], z["targets"], z["in_len"], z["tgt_len"])
    np.testing.assert_allclose(loss, z["loss"], rtol=2e-6)
    np.testing.assert_allclose(dlogits, z["dlogits"], atol=5e-6)  # fp32 log-space roundoff
    # exactly zero beyond input length
    for b in range(loss.shape[0]):
        assert np.all(dlogits[b, int(z["in_len"][b]):] == 0)
    zi = load(golden_dir, "ctc_inf.npz")
    li, _ = O.ctc_loss_fwd_bwd(zi["logits"], np.array([[4, 4, 4]]), np.array([3]), np.array([3]), want_grad=False)
    assert np.isinf(li[0]) and np.isinf(zi["loss"][0])


@pytest.mark.parametrize("name", ["train_step.npz", "train_step_patch.npz"])
def test_train_step(golden_dir, name):
    """Four steps of the reference's step body (rnn_trainer.py:527-558); train_step_patch.npz: patch_size 14 / stride 4
    (rnn_model.py:106-119), i.e. the unfold's adjoint and layer 0's weight gradient over overlapping rows."""
    z = load(golden_dir, name)
    F, H, D, C, L, ps, st = [int(v) for v in z["cfg"]]
    sd = sd_of(z, "sd0::")
    clip = float(z["clip"]); warm = int(z["warmup"])
    m = {k: np.zeros_like(v) for k, v in sd.items()}
    v = {k: np.zeros_like(v_) for k, v_ in sd.items()}
    steps = {k: 0 for k in sd}
    lr_cfg = dict(bias=(0.005, 0.0001, 0.0), day=(0.005, 0.0001, 0.0), other=(0.005, 0.0001, 0.001))
    for it in range(4):
        feats, n = O.transform_data(z["x"], z["n_time_steps"], "val")
        if it == 0:
            np.testing.assert_allclose(feats, z["feats0"], atol=2e-6)
        loss, loss_b, logits, g = O.model_loss_and_grads(sd, feats, z["day_idx"], z["targets"], O.adjusted_lens(n, ps, st),
                                                         z["tgt_len"], L, ps, st)
        np.testing.assert_allclose(loss, z[f"loss{it}"], rtol=1e-5)
        if it == 0:
            np.testing.assert_allclose(logits, z["logits0"], atol=2e-5)
            gold = sd_of(z, "grad0::")
            assert set(gold) == set(g), (sorted(set(gold) ^ set(g)))
            for k in gold:
                scale = max(1e-6, float(np.abs(gold[k]).max()))
                np.testing.assert_allclose(g[k], gold[k], atol=2e-4 * scale + 1e-7, err_msg=k)
        norm, gc = O.clip_grad_norm(g, clip)
        np.testing.assert_allclose(norm, z[f"gnorm{it}"], rtol=2e-5)
        lrs = z[f"lr{it}"]
        for k in gc:
            grp = O.param_group_of(k)
            lr_max, lr_min, wd = lr_cfg[grp]
            fac = O.lr_factor(it, lr_min, lr_max, 120000, warm)
            lr = lr_max * fac
            gi = {"bias": 0, "day": 1, "other": 2}[grp]
            assert abs(lr - lrs[gi]) < 1e-12
            steps[k] += 1
            sd[k], m[k], v[k] = O.adamw_step(sd[k], gc[k], m[k], v[k], steps[k], lr, wd)
        gold_sd = sd_of(z, f"sd{it+1}::")
        for k in gold_sd:
            np.testing.assert_allclose(sd[k], gold_sd[k], atol=3e-6, err_msg=f"step{it} {k}")


@pytest.mark.parametrize("name", ["train_step.npz", "train_step_patch.npz"])
def test_torch_cpu_step_is_the_reference_step(golden_dir, name):
    """oracle/torch_cpu_step.py (bench.py's cpu_baseline) reproduces the reference's four steps: losses, gradient norms,
    learning rates and every parameter after each AdamW step."""
    import torch
    from oracle import torch_cpu_step as TC
    z = load(golden_dir, name)
    F, H, D, C, L, ps, st = [int(v) for v in z["cfg"]]
    torch.set_num_threads(4)
    m = TC.CpuGRUDecoder(F, H, D, C, L, ps, st)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd_of(z, "sd0::").items()})
    w = int(z["warmup"])
    args = dict(lr_max=0.005, lr_min=0.0001, lr_decay_steps=120000, lr_warmup_steps=w, lr_max_day=0.005, lr_min_day=0.0001,
                lr_decay_steps_day=120000, lr_warmup_steps_day=w, beta0=0.9, beta1=0.999, epsilon=0.1, weight_decay=0.001,
                weight_decay_day=0, grad_norm_clip_value=float(z["clip"]))
    tr = TC.CpuTrainer(m, args)
    t = lambda k: torch.from_numpy(z[k])
    for it in range(4):
        np.testing.assert_allclose([g["lr"] for g in tr.opt.param_groups], z[f"lr{it}"], rtol=1e-12)
        loss, gn = tr.step(t("x"), t("day_idx"), t("targets"), t("n_time_steps"), t("tgt_len"))
        np.testing.assert_allclose(loss, z[f"loss{it}"], rtol=1e-6)
        np.testing.assert_allclose(gn, z[f"gnorm{it}"], rtol=1e-5)
        for k, ref in sd_of(z, f"sd{it+1}::").items():
            np.testing.assert_allclose(m.state_dict()[k].numpy(), ref, atol=1e-6, err_msg=f"step{it} {k}")


def test_lr_table(golden_dir):
    z = load(golden_dir, "lr_table.npz")
    for s, fac in zip(z["steps"], z["factors"]):
        f = O.lr_factor(int(s), 0.0001, 0.005, 120000, 1000)
        assert abs(f - fac[0]) < 1e-15 and abs(f - fac[2]) < 1e-15
    assert O.lr_factor(0, 0.0001, 0.005, 120000, 1000) == 0.0
    np.testing.assert_allclose(z["first_lrs"][:, 0], [0.0, 5e-6, 1e-5], rtol=1e-12)
    # lr_scheduler_type 'linear' (torch LinearLR, rnn_trainer.py:228-234): the closed form equals torch's chained updates
    tot = int(z["linear_total"])
    for i, lrs in enumerate(z["linear_lrs"]):
        f = O.linear_lr_factor(i, 0.0001, 0.005, tot)
        np.testing.assert_allclose(lrs, [0.005 * f, 0.005 * f, 0.005 * f], rtol=1e-12)
    assert abs(z["linear_lrs"][-1][0] - 0.0001) < 1e-15


def test_transform(golden_dir):
    z = load(golden_dir, "transform.npz")
    for cut in (0, 1, 2):
        y, n = O.transform_data(z["x"], z["n_time_steps"], "train", white_noise=z["white"], white_noise_std=1.0,
                                offset_noise=z["offset"], constant_offset_std=0.2, cut=cut)
        np.testing.assert_allclose(y, z[f"train_cut{cut}"], atol=3e-6)
        np.testing.assert_array_equal(n, z[f"train_cut{cut}_n"])
    y, n = O.transform_data(z["x"], z["n_time_steps"], "val")
    np.testing.assert_allclose(y, z["val"], atol=3e-6)
    np.testing.assert_array_equal(n, z["val_n"])
    for tag, axis in (("last", -1), ("time", 1)):   # static gain + random walk on too (rnn_trainer.py:449-453,464-465)
        y, n = O.transform_data(z["x"], z["n_time_steps"], "train", white_noise=z["white"], white_noise_std=1.0,
                                offset_noise=z["offset"], constant_offset_std=0.2, cut=1,
                                static_gain_noise=z[f"full_{tag}_sg"], static_gain_std=0.1,
                                random_walk_noise=z[f"full_{tag}_rw"], random_walk_std=0.05, random_walk_axis=axis)
        np.testing.assert_allclose(y, z[f"full_{tag}"], atol=5e-6)


def test_greedy_and_edit(golden_dir):
    z = load(golden_dir, "greedy.npz")
    for b in range(z["logits"].shape[0]):
        d = O.greedy_decode_trainer(z["logits"][b], int(z["lens"][b]))
        np.testing.assert_array_equal(d, z[f"trainer_{b}"])
        e = O.edit_distance(d, z["labels"][b][:int(z["lab_len"][b])])
        assert e == int(z[f"edit_{b}"])
        np.testing.assert_array_equal(O.greedy_decode_evaluate(z["logits"][b]), z[f"evaluate_{b}"])
    np.testing.assert_array_equal(O.rearrange_speech_logits(z["logits"]), z["rearranged"])


def test_evalstep(golden_dir):
    z = load(golden_dir, "evalstep.npz")
    F, H, D, C, L, ps, st = [int(v) for v in z["cfg"]]
    x = O.gauss_smooth(z["x"], 2, 100, "valid")
    logits, _ = O.model_fwd(sd_of(z), x, np.array([int(z["day"])]), L, ps, st)
    assert logits.shape == z["logits"].shape            # T'' = floor((T-8-14)/4)+1
    np.testing.assert_allclose(logits, z["logits"], atol=2e-5)


def test_prefix_beam_known_answer():
    """ctc_prefix_beam_search_test.cc:18-59 — the reference's only numeric known answer."""
    p = np.array([[0.25, 0.40, 0.35], [0.40, 0.35, 0.25], [0.10, 0.50, 0.40]], dtype=np.float32)
    res = O.prefix_beam_search(np.log(p), first_beam=3, second_beam=3)
    assert [r[0] for r in res] == [(2, 1), (1, 2), (1,)]
    np.testing.assert_allclose([math.exp(r[1]) for r in res], [0.2185, 0.1550, 0.1525], rtol=1e-5)
    np.testing.assert_allclose([math.exp(r[2]) for r in res], [0.07, 0.064, 0.07], rtol=1e-5)
    assert [r[3] for r in res] == [[0, 2], [0, 2], [2]]


def test_lm_prologue():
    rng = np.random.default_rng(0)
    lg = rng.standard_normal((7, 41)).astype(np.float32)
    lp = O.lm_prologue(lg, np.zeros_like(lg), math.log(90.0))
    ref = lg - np.log(np.exp(lg).sum(-1, keepdims=True))
    ref[:, 0] -= math.log(90.0)
    np.testing.assert_allclose(lp, ref, atol=1e-5)
