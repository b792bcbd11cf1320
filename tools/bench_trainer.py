#!/usr/bin/env python3
"""Trainer-level throughput: BrainToTextDecoder_Trainer.train() (rnn_trainer.py:486-651 counterpart) at the C2 shape on a
synthetic device-resident dataset -- the whole loop incl. batch assembly, augmentation, the lagged loss read and logging --
against bench.py's bare step.  Two runs of different length; the difference cancels construction and warm-up."""
import json, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "nejm-brain-to-text_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch


def make_args(tmp, n_batches, amp=False, shipped=False):
    sessions = [f"t15.2023.{8 + d // 28:02d}.{1 + d % 28:02d}" for d in range(45)]
    # shipped: the model block of the reference's rnn_args.yaml (model_training/rnn_args.yaml: n_units 768, patch 14 / 4, rnn_dropout 0.4,
    # input_layer_dropout 0.2) -- BASELINE configs[2]'s shape -- instead of configs[1]'s
    return {
        'model': {'n_input_features': 512, 'n_units': 768 if shipped else 512, 'rnn_dropout': 0.4 if shipped else 0.0, 'rnn_trainable': True, 'n_layers': 5,
                  'patch_size': 14 if shipped else 0, 'patch_stride': 4 if shipped else 0,
                  'input_network': {'n_input_layers': 1, 'input_layer_sizes': [512], 'input_trainable': True, 'input_layer_dropout': 0.2 if shipped else 0.0}},
        'gpu_number': '0', 'mode': 'train', 'use_amp': amp,
        'output_dir': os.path.join(tmp, 'out'), 'checkpoint_dir': os.path.join(tmp, 'out', 'checkpoint'),
        'init_from_checkpoint': False, 'init_checkpoint_path': None, 'save_best_checkpoint': False,
        'save_all_val_steps': False, 'save_final_model': False, 'save_val_metrics': False, 'early_stopping': False,
        'early_stopping_val_steps': 20, 'num_training_batches': n_batches, 'lr_scheduler_type': 'cosine',
        'lr_max': 0.005, 'lr_min': 0.0001, 'lr_decay_steps': 120000, 'lr_warmup_steps': 1000, 'lr_max_day': 0.005,
        'lr_min_day': 0.0001, 'lr_decay_steps_day': 120000, 'lr_warmup_steps_day': 1000, 'beta0': 0.9, 'beta1': 0.999,
        'epsilon': 0.1, 'weight_decay': 0.001, 'weight_decay_day': 0, 'seed': 10, 'grad_norm_clip_value': 10,
        'batches_per_train_log': 200, 'batches_per_val_step': 10 ** 9, 'batches_per_save': 0,
        'log_individual_day_val_PER': True, 'log_val_skip_logs': True, 'save_val_logits': False, 'save_val_data': False,
        'dataset': {'data_transforms': {'white_noise_std': 1.0, 'constant_offset_std': 0.2, 'random_walk_std': 0.0,
                                        'random_walk_axis': -1, 'static_gain_std': 0.0, 'random_cut': 3,
                                        'smooth_kernel_size': 100, 'smooth_data': True, 'smooth_kernel_std': 2},
                    'neural_dim': 512, 'batch_size': 64, 'n_classes': 41, 'max_seq_elements': 500, 'days_per_batch': 4,
                    'seed': 1, 'num_dataloader_workers': 0, 'loader_shuffle': False, 'must_include_days': None,
                    'test_percentage': 0.1, 'feature_subset': None, 'dataset_dir': '/nonexistent', 'bad_trials_dict': None,
                    'sessions': sessions, 'dataset_probability_val': [1] * len(sessions), 'device_resident': True,
                    'synthetic': {'max_T': 500, 'min_T': 500, 'max_S': 60, 'val_batches': 1}},
    }


def run(n, amp=None, shipped=False):
    from rnn_trainer import BrainToTextDecoder_Trainer
    with tempfile.TemporaryDirectory() as tmp:
        amp = bool(int(os.environ.get("B2T_TRAINER_AMP", "0"))) if amp is None else amp
        tr = BrainToTextDecoder_Trainer(make_args(tmp, n, amp=amp, shipped=shipped))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        stats = tr.train()
        torch.cuda.synchronize()
        return time.perf_counter() - t0, stats


def main():
    n0, n1 = 40, int(os.environ.get("B2T_TRAINER_STEPS", "160"))
    run(12)                                   # first touch: library load, queue calibration, allocator
    t0, _ = run(n0)
    t1, st = run(n1)
    ms = (t1 - t0) / (n1 - n0) * 1e3
    print(json.dumps({"trainer_ms_per_step": round(ms, 3), "trainer_sentences_per_s": round(64e3 / ms, 1),
                      "runs": {str(n0): round(t0, 3), str(n1): round(t1, 3)}, "final_loss": float(st['train_losses'][-1]),
                      "workload": "rnn_trainer.train(): 5-layer GRU-512, B=64, T=500 (all trials 500 frames), 45 sessions, 4 days per batch, device-resident synthetic dataset, augmentation on, no validation inside the window"}))


if __name__ == "__main__":
    main()
