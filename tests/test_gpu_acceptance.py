"""The north star's acceptance ("PER within +-0.1 % of the pretrained t15 baseline") as a runnable check.

* test_t15_checkpoint_acceptance: tools/accept_t15.py on the REAL checkpoint and validation split -- skipped when
  data/t15_pretrained_rnn_baseline, data/hdf5_data_final or h5py are absent (they are not in this image).
* test_acceptance_tool_mechanics: the same tool end to end on a checkpoint THIS trainer wrote for the synthetic copy task
  (checkpoint dir layout of the reference: checkpoint/{best_checkpoint,args.yaml}, `_orig_mod.` key prefix, stored val_PER):
  the fp32 validation reproduces the stored PER exactly, the bf16 mode stays within the stated tolerance."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nejm-brain-to-text_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def test_t15_checkpoint_acceptance():
    import accept_t15
    model_path = os.environ.get("B2T_T15_MODEL", os.path.join(ROOT, "data", "t15_pretrained_rnn_baseline"))
    data_dir = os.environ.get("B2T_T15_DATA", os.path.join(ROOT, "data", "hdf5_data_final"))
    ok, why = accept_t15.available(model_path, data_dir)
    if not ok:
        pytest.skip(f"real-data acceptance not runnable here: {why}")
    ok, out = accept_t15.run(model_path, data_dir, tol=1e-3)
    assert ok, out


def test_acceptance_tool_mechanics(tmp_path):
    import accept_t15
    import b2t_ops as ops
    from rnn_trainer import BrainToTextDecoder_Trainer
    from test_gpu_trainer import make_args
    was = ops.AMP["on"]
    try:
        args = make_args(str(tmp_path / "run"), n_batches=60, patch=(4, 2), dropout=(0.0, 0.0))
        args['batches_per_val_step'] = 20
        tr = BrainToTextDecoder_Trainer(args)
        st = tr.train()
        ck_dir = args['checkpoint_dir']
        assert os.path.isfile(os.path.join(ck_dir, "best_checkpoint"))
        ck = torch.load(os.path.join(ck_dir, "best_checkpoint"), weights_only=False, map_location="cpu")
        assert all(k.startswith("_orig_mod.") for k in ck["model_state_dict"]) and "val_PER" in ck
        assert set(accept_t15.strip_prefixes({"module._orig_mod.gru.weight_hh_l0": 1})) == {"gru.weight_hh_l0"}
        assert os.path.isfile(os.path.join(ck_dir, "args.yaml"))      # written by save_model_checkpoint, as the reference does
        ok, out = accept_t15.run(str(tmp_path / "run" / "out"), "/nonexistent", tol=0.02)
        assert out["checkpoint_val_PER"] == pytest.approx(min(st["val_PERs"]), abs=1e-9) or out["checkpoint_val_PER"] in st["val_PERs"]
        assert abs(out["PER_fp32"] - out["checkpoint_val_PER"]) < 1e-9, out      # same weights, same split, same arithmetic
        assert ok, out
    finally:
        ops.set_amp(was)
