"""End-to-end trainer on the GPU with the synthetic batch source: the reference's train() contract
(return dict, log/ckpt files, checkpoint keys) and that training actually learns."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def make_args(tmp, n_batches=40, patch=(4, 2), dropout=(0.1, 0.1)):
    sessions = [f"t15.2023.08.{d:02d}" for d in range(11, 17)]
    return {
        'model': {'n_input_features': 32, 'n_units': 64, 'rnn_dropout': dropout[0], 'rnn_trainable': True, 'n_layers': 2,
                  'patch_size': patch[0], 'patch_stride': patch[1],
                  'input_network': {'n_input_layers': 1, 'input_layer_sizes': [32], 'input_trainable': True,
                                    'input_layer_dropout': dropout[1]}},
        'gpu_number': '0', 'mode': 'train', 'use_amp': True, 'amd_bf16_matmul': False,   # exact fp32: these tests compare with the fp32 oracle
        'output_dir': os.path.join(tmp, 'out'), 'checkpoint_dir': os.path.join(tmp, 'out', 'checkpoint'),
        'init_from_checkpoint': False, 'init_checkpoint_path': None, 'save_best_checkpoint': True,
        'save_all_val_steps': False, 'save_final_model': False, 'save_val_metrics': True, 'early_stopping': False,
        'early_stopping_val_steps': 20, 'num_training_batches': n_batches, 'lr_scheduler_type': 'cosine',
        'lr_max': 0.02, 'lr_min': 0.001, 'lr_decay_steps': n_batches, 'lr_warmup_steps': 5, 'lr_max_day': 0.02,
        'lr_min_day': 0.001, 'lr_decay_steps_day': n_batches, 'lr_warmup_steps_day': 5, 'beta0': 0.9, 'beta1': 0.999,
        'epsilon': 0.1, 'weight_decay': 0.001, 'weight_decay_day': 0, 'seed': 10, 'grad_norm_clip_value': 10,
        'batches_per_train_log': 10, 'batches_per_val_step': 20, 'batches_per_save': 0,
        'log_individual_day_val_PER': True, 'log_val_skip_logs': False, 'save_val_logits': True, 'save_val_data': False,
        'dataset': {'data_transforms': {'white_noise_std': 0.2, 'constant_offset_std': 0.05, 'random_walk_std': 0.0,
                                        'random_walk_axis': -1, 'static_gain_std': 0.0, 'random_cut': 3,
                                        'smooth_kernel_size': 100, 'smooth_data': True, 'smooth_kernel_std': 2},
                    'neural_dim': 32, 'batch_size': 16, 'n_classes': 41, 'max_seq_elements': 500, 'days_per_batch': 3,
                    'seed': 1, 'num_dataloader_workers': 0, 'loader_shuffle': False, 'must_include_days': None,
                    'test_percentage': 0.1, 'feature_subset': None, 'dataset_dir': '/nonexistent', 'bad_trials_dict': None,
                    'sessions': sessions, 'dataset_probability_val': [1] * len(sessions),
                    'synthetic': {'max_T': 80, 'min_T': 50, 'max_S': 6, 'val_batches': 3}},
    }


def test_trainer_end_to_end(tmp_path):
    from rnn_trainer import BrainToTextDecoder_Trainer
    args = make_args(str(tmp_path))
    tr = BrainToTextDecoder_Trainer(args)
    stats = tr.train()
    assert set(stats) == {'train_losses', 'val_losses', 'val_PERs', 'val_metrics'}
    assert len(stats['train_losses']) == 40 and len(stats['val_PERs']) == 3       # batches 0, 20, 39
    first, last = np.mean(stats['train_losses'][:5]), np.mean(stats['train_losses'][-5:])
    assert np.all(np.isfinite(stats['train_losses'])) and last < 0.8 * first, (first, last)
    vm = stats['val_metrics'][-1]
    for k in ('decoded_seqs', 'true_seq', 'phone_seq_lens', 'transcription', 'losses', 'block_nums', 'trial_nums',
              'day_indicies', 'day_PERs', 'avg_PER', 'avg_loss', 'logits', 'n_time_steps'):
        assert k in vm, k
    assert 0.0 <= vm['avg_PER'] <= 2.0
    # files the reference writes
    assert os.path.exists(os.path.join(args['output_dir'], 'training_log'))
    ck = os.path.join(args['checkpoint_dir'], 'best_checkpoint')
    assert os.path.exists(ck) and os.path.exists(os.path.join(args['checkpoint_dir'], 'args.yaml'))
    assert os.path.exists(os.path.join(args['checkpoint_dir'], 'val_metrics.pkl'))
    c = torch.load(ck, weights_only=False)
    assert set(c) == {'model_state_dict', 'optimizer_state_dict', 'scheduler_state_dict', 'val_PER', 'val_loss'}
    assert all(k.startswith('_orig_mod.') for k in c['model_state_dict'])
    assert [g['group_type'] for g in c['optimizer_state_dict']['param_groups']] == ['bias', 'day_layer', 'other']
    # evaluate_model.py-style load: strip prefixes, load, one decoding step on a raw trial
    from rnn_model import GRUDecoder
    from evaluate_model_helpers import runSingleDecodingStep
    m = GRUDecoder(32, 64, 6, 41, 0.1, 0.1, 2, 4, 2)
    sd = {k.replace("module.", "").replace("_orig_mod.", ""): v for k, v in c['model_state_dict'].items()}
    m.load_state_dict(sd)
    m.to("cuda:0").eval()
    x = torch.randn(1, 70, 32).to(torch.bfloat16)
    lg = runSingleDecodingStep(x, 2, m, {'dataset': {'data_transforms': args['dataset']['data_transforms']}}, "cuda:0")
    assert lg.shape == (1, (70 - 8 - 4) // 2 + 1, 41) and lg.dtype == np.float32


def test_trainer_resume_and_validation_matches_oracle(tmp_path):
    """validation(): PER and loss equal the oracle's greedy/edit-distance/CTC on the same logits."""
    from oracle import b2t_oracle as O
    from rnn_trainer import BrainToTextDecoder_Trainer
    args = make_args(str(tmp_path), n_batches=3, patch=(0, 0), dropout=(0.0, 0.0))
    args['batches_per_val_step'] = 100
    tr = BrainToTextDecoder_Trainer(args)
    vm = tr.validation(tr.val_loader, return_logits=True)
    tot_e, tot_l, losses = 0, 0, []
    for bi, batch in enumerate(tr.val_loader):
        logits = vm['logits'][bi]; lens = vm['n_time_steps'][bi]
        for b in range(logits.shape[0]):
            d = O.greedy_decode_trainer(logits[b], int(lens[b]))
            np.testing.assert_array_equal(d, vm['decoded_seqs'][bi][b])
            S = int(batch['phone_seq_lens'][b])
            tot_e += O.edit_distance(d, batch['seq_class_ids'][b][:S].numpy()); tot_l += S
        lo, _ = O.ctc_loss_fwd_bwd(logits, batch['seq_class_ids'].numpy(), lens, batch['phone_seq_lens'].numpy(), want_grad=False)
        losses.append(lo.mean())
    assert abs(vm['avg_PER'] - tot_e / tot_l) < 1e-9
    np.testing.assert_allclose(vm['avg_loss'], np.mean(np.repeat(losses, 2)), rtol=1e-5)


def test_resident_dataset_matches_source_batches(tmp_path):
    """the device-resident flat dataset reproduces the source's batch dicts exactly (features, labels,
    lengths, days), batch by batch and for arbitrary row selections; the flat-binary round trip is lossless."""
    from dataset import SyntheticTrials, ResidentDataset
    src = SyntheticTrials(n_batches=5, batch_size=12, n_days=4, n_features=36, n_classes=41, days_per_batch=2, max_T=90,
                          min_T=40, max_S=9, seed=3)
    rd = ResidentDataset.from_batches(src, device='cuda:0')
    assert len(rd) == 5 and rd.n_trials == 60
    for i in range(len(src)):
        a, b = src[i], rd.batch_of(i)
        for k in ('input_features', 'seq_class_ids', 'n_time_steps', 'phone_seq_lens', 'day_indicies', 'block_nums', 'trial_nums'):
            np.testing.assert_array_equal(b[k].cpu().numpy(), a[k].numpy(), err_msg=f"{k} batch {i}")
        assert b['input_features'].dtype == torch.float32 and b['seq_class_ids'].dtype == torch.int64
    # arbitrary rows (with repeats), an odd feature width (scalar copy path) and the file round trip
    rows = torch.tensor([59, 0, 17, 17, 33])
    got = rd.batch(rows)
    T = int(got['n_time_steps'].max())
    for j, r in enumerate(rows.tolist()):
        n = int(rd.host['n_time_steps'][r]); o = int(rd.host['feat_off'][r])
        np.testing.assert_array_equal(got['input_features'][j, :n].cpu().numpy(), rd.host['feat'][o:o + n])
        assert float(got['input_features'][j, n:].abs().sum()) == 0.0
    assert got['input_features'].shape == (5, T, 36)
    p = str(tmp_path / "flat.npz")
    rd.save(p)
    rd2 = ResidentDataset.load(p, device='cuda:0')
    b1, b2 = rd.batch_of(2), rd2.batch_of(2)
    assert torch.equal(b1['input_features'], b2['input_features']) and torch.equal(b1['seq_class_ids'], b2['seq_class_ids'])
    src7 = SyntheticTrials(n_batches=1, batch_size=5, n_days=2, n_features=7, n_classes=41, days_per_batch=1, max_T=30, min_T=20,
                           max_S=4, seed=1)
    rd7 = ResidentDataset.from_batches(src7, device='cuda:0')
    np.testing.assert_array_equal(rd7.batch_of(0)['input_features'].cpu().numpy(), src7[0]['input_features'].numpy())


def test_resident_cache_is_tied_to_its_source(tmp_path):
    """The flat-binary cache carries a fingerprint of what it was built from: a changed source (seed, batch count, ...) or a
    torn / foreign file means REBUILD -- never a silently stale batch index, never a crash; only the writer process writes,
    atomically (advisor, round 3)."""
    import warnings
    from dataset import SyntheticTrials, ResidentDataset
    kw = dict(batch_size=6, n_days=3, n_features=16, n_classes=41, days_per_batch=2, max_T=40, min_T=20, max_S=5)
    a, b = SyntheticTrials(n_batches=3, seed=3, **kw), SyntheticTrials(n_batches=3, seed=4, **kw)
    assert ResidentDataset.source_fingerprint(a) == ResidentDataset.source_fingerprint(SyntheticTrials(n_batches=3, seed=3, **kw))
    assert ResidentDataset.source_fingerprint(a) != ResidentDataset.source_fingerprint(b)
    assert ResidentDataset.source_fingerprint(a) != ResidentDataset.source_fingerprint(SyntheticTrials(n_batches=4, seed=3, **kw))
    p = str(tmp_path / "cache.npz")
    ra = ResidentDataset.load_or_build(p, a, 'cuda:0', writer=False)
    assert not os.path.exists(p), "a non-writer rank must not write the cache"
    ra = ResidentDataset.load_or_build(p, a, 'cuda:0')
    assert os.path.exists(p) and not [f for f in os.listdir(tmp_path) if ".tmp." in f]
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        ra2 = ResidentDataset.load_or_build(p, a, 'cuda:0')          # same source: served from the file, silently
    assert torch.equal(ra.batch_of(1)['input_features'], ra2.batch_of(1)['input_features'])
    with pytest.warns(UserWarning, match="different source"):
        rb = ResidentDataset.load_or_build(p, b, 'cuda:0')           # other seed: rebuilt, and the file now belongs to b
    np.testing.assert_array_equal(rb.batch_of(1)['input_features'].cpu().numpy(), b[1]['input_features'].numpy())
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        ResidentDataset.load_or_build(p, b, 'cuda:0')
    with open(p, "r+b") as f:                                        # a torn file: rebuilt, not a BadZipFile crash
        f.truncate(os.path.getsize(p) // 2)
    with pytest.warns(UserWarning, match="rebuilding"):
        rb2 = ResidentDataset.load_or_build(p, b, 'cuda:0')
    assert torch.equal(rb.batch_of(2)['input_features'], rb2.batch_of(2)['input_features'])
    ResidentDataset.load(p, 'cuda:0', fingerprint=ResidentDataset.source_fingerprint(b))   # and the rewritten file is whole again


def test_trainer_with_device_resident_dataset(tmp_path):
    """The trainer fed from the device-resident dataset produces the same losses and validation metrics as with the
    DataLoader path (same batch composition and order, same augmentation draws)."""
    from rnn_trainer import BrainToTextDecoder_Trainer
    out = {}
    for resident in (False, True):
        args = make_args(str(tmp_path / ("r" if resident else "l")), n_batches=21)
        args['dataset']['device_resident'] = resident
        tr = BrainToTextDecoder_Trainer(args)
        st = tr.train()
        out[resident] = (st['train_losses'], st['val_PERs'])
    np.testing.assert_allclose(out[True][0], out[False][0], rtol=1e-6)
    np.testing.assert_allclose(out[True][1], out[False][1], rtol=1e-9)


def _per_args(tmp, amp, n_batches):
    a = make_args(tmp, n_batches=n_batches, patch=(4, 2), dropout=(0.0, 0.0))
    a['model'].update(n_input_features=64, n_units=128, n_layers=3)
    a['model']['input_network']['input_layer_sizes'] = [64]
    a['dataset'].update(neural_dim=64, batch_size=32)
    a['dataset']['synthetic'] = {'max_T': 120, 'min_T': 60, 'max_S': 12, 'val_batches': 40}      # 1280 val sentences, ~9.6 k phonemes
    a['batches_per_val_step'] = 10 ** 9
    a['batches_per_train_log'] = 100
    a['save_best_checkpoint'] = False; a['save_val_metrics'] = False; a['save_val_logits'] = False
    a['lr_max'] = a['lr_max_day'] = 0.02
    a['amd_bf16_matmul'] = bool(amp)
    return a


def test_bf16_mode_phoneme_error_rate_same_weights_0p1_percent_trained_0p3_percent(tmp_path):
    """The acceptance BASELINE.json's north star names for the bf16 regime (`use_amp: true`, rnn_trainer.py:535 autocast):
    phoneme error rate within +-0.1 % (absolute) of the fp32 path.  Two ways, on the learnable synthetic copy task with a
    validation set of 1280 sentences (~9.6 k phonemes: one phoneme = 0.01 %):
      (a) the SAME fp32-trained weights decoded through the fp32 and through the bf16-operand forward (what evaluating the
          pretrained t15 checkpoint in the other precision means);
      (b) the model TRAINED from the same seed and data in fp32 and in bf16 mode, each validated in its own precision.
    (a) is the acceptance itself and holds to 0.1 %.  (b) compares two training TRAJECTORIES, and 1600 steps of SGD amplify any
    difference in rounding: two exact-fp32 runs that differ only in the order of fp32 accumulations end 0.06 % apart (PER 10.697 %
    with the fused projections, 10.760 % without: tools/r4_per_noise.py), and the bf16 run has landed 0.02 % and 0.21 % from the fp32
    one in this round depending on the summation order of the day-bias column sums (10.906 % now) -- so (b) is held to 0.3 %
    (a few fp32-vs-fp32 distances), and (a), which has no trajectory in it, stays the sharp check."""
    import b2t_ops as ops
    from rnn_trainer import BrainToTextDecoder_Trainer
    # to the plateau of this task (PER ~10.7 %: the templates overlap); measured |difference| 0.02 % there, 0.14 % at 1200.  The host-bound
    # trainer loop makes this the suite's longest test (100-200 s by box): 1200 steps by default, the full 1600 with B2T_TEST_FULL=1
    N_STEPS = 1600 if os.environ.get("B2T_TEST_FULL") else 1200
    was = ops.AMP["on"]
    try:
        res = {}
        for amp in (False, True):
            tr = BrainToTextDecoder_Trainer(_per_args(str(tmp_path / ("amp" if amp else "f32")), amp, N_STEPS))
            assert ops.AMP["on"] == amp
            st = tr.train()
            res[amp] = dict(PER=st['val_PERs'][-1], loss=st['val_losses'][-1], trainer=tr,
                            n_phonemes=sum(int(np.sum(p)) for p in st['val_metrics'][-1]['phone_seq_lens']))
        f32, amp = res[False], res[True]
        assert f32['n_phonemes'] > 8000
        assert f32['PER'] < 0.12, f"the copy task did not train: PER {f32['PER']:.4f}"
        # (b) trained in each precision
        assert abs(amp['PER'] - f32['PER']) <= 3e-3, (amp['PER'], f32['PER'])
        # (a) the fp32-trained weights through the other forward
        tr = f32['trainer']
        ops.set_amp(True)
        per_other = tr.validation(tr.val_loader)['avg_PER']
        ops.set_amp(False)
        per_same = tr.validation(tr.val_loader)['avg_PER']
        assert abs(per_same - f32['PER']) < 1e-12
        assert abs(per_other - per_same) <= 1e-3, (per_other, per_same)
        print(f"PER fp32-trained {f32['PER']:.4%} (bf16 forward of the same weights {per_other:.4%}); bf16-trained {amp['PER']:.4%}; "
              f"{f32['n_phonemes']} validation phonemes")
    finally:
        ops.set_amp(was)
