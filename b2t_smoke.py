"""smoke(): one tiny training step of the hot path on cuda:0 (augment+smooth -> day layer -> GRU x2 ->
head -> CTC -> backward -> clip -> AdamW), checked against the oracle (test infrastructure)."""
import numpy as np
import torch


def run():
    assert torch.cuda.is_available(), "smoke() needs the MI355X"
    from rnn_model import GRUDecoder
    from b2t_train_step import TrainStep
    import b2t_ops as ops
    from oracle import b2t_oracle as O

    dev = torch.device("cuda:0")
    torch.manual_seed(10)
    F, H, D, C, L, B, T, S = 32, 64, 3, 41, 2, 4, 24, 5
    model = GRUDecoder(F, H, D, C, 0.0, 0.0, L, 0, 0)
    sd0 = {k: v.detach().numpy().copy() for k, v in model.state_dict().items()}
    model = model.to(dev).train()
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, T, F, generator=g)
    day = torch.tensor([0, 2, 2, 0])
    tgt = torch.randint(1, C, (B, S), generator=g)
    tl = torch.tensor([5, 3, 4, 2]); nt = torch.tensor([24, 20, 24, 15])
    for b in range(B):
        tgt[b, tl[b]:] = 0
    args = dict(lr_max=0.005, lr_min=0.0001, lr_decay_steps=1000, lr_warmup_steps=2, lr_max_day=0.005,
                lr_min_day=0.0001, lr_decay_steps_day=1000, lr_warmup_steps_day=2, beta0=0.9, beta1=0.999,
                epsilon=0.1, weight_decay=0.001, weight_decay_day=0, grad_norm_clip_value=10,
                _debug_keep_unclipped=True)
    ts = TrainStep(model, args)
    feats = ops.augment_smooth(x.to(dev), 2, 100, "same")
    loss, gnorm = ts.step(feats, day, tgt, nt, tl)
    torch.cuda.synchronize()
    fo, no = O.transform_data(x.numpy(), nt.numpy(), "val")
    lo, _, _, go = O.model_loss_and_grads(sd0, fo, day.numpy(), tgt.numpy(), no, tl.numpy(), L)
    assert abs(float(loss) - float(lo)) <= 1e-4 * abs(float(lo)), (float(loss), float(lo))
    got = ts.last_unclipped_grads()
    for k, ref in go.items():
        err = np.abs(got[k] - ref).max()
        assert err <= 1e-3 * max(1e-6, np.abs(ref).max()), (k, err)
    norm_o, _ = O.clip_grad_norm(go, 10)
    assert abs(float(gnorm) - float(norm_o)) <= 1e-4 * float(norm_o)
    print(f"smoke OK: loss {float(loss):.5f} (oracle {float(lo):.5f}), grad norm {float(gnorm):.5f}")
