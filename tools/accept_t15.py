#!/usr/bin/env python3
"""Acceptance on the real checkpoint (north star: "phoneme-error-rate within +-0.1 % of the pretrained t15 baseline").

Loads data/t15_pretrained_rnn_baseline/checkpoint/{args.yaml, best_checkpoint} (keys with the reference's `_orig_mod.` /
`module.` prefixes, evaluate_model.py:72-77), runs the validation split of data/hdf5_data_final through THIS package's
`BrainToTextDecoder_Trainer.validation()` (rnn_trainer.py:659-768 counterpart: smoothing, forward, greedy decode, edit
distance) in exact fp32 and in the bf16 mode, and compares the aggregate PER with the value the reference's trainer stored
in the checkpoint (`checkpoint['val_PER']`, rnn_trainer.py:392-398) to +-0.1 % absolute.

Nothing of /root/reference is imported.  The session files are HDF5: `h5py` must be importable where the data lives
(it is not part of this image, and neither is the data: tests/test_gpu_acceptance.py skips when either is absent).

  python tools/accept_t15.py --model_path data/t15_pretrained_rnn_baseline --data_dir data/hdf5_data_final
exit code 0 = both precisions within tolerance; prints one JSON line."""
import argparse
import copy
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nejm-brain-to-text_amd"))


def available(model_path, data_dir):
    """(ok, why not)"""
    ck = os.path.join(model_path, "checkpoint")
    if not (os.path.isfile(os.path.join(ck, "best_checkpoint")) and os.path.isfile(os.path.join(ck, "args.yaml"))):
        return False, f"{ck}/{{best_checkpoint,args.yaml}} not found"
    if not os.path.isdir(data_dir):
        return False, f"{data_dir} not found"
    try:
        import h5py  # noqa: F401
    except ImportError:
        return False, "h5py is not importable (the session files are HDF5)"
    return True, ""


def strip_prefixes(sd):
    """evaluate_model.py:74-76: the trainer saved a torch.compile'd (and possibly DataParallel) module."""
    out = {}
    for k, v in sd.items():
        out[k.replace("module.", "").replace("_orig_mod.", "")] = v
    return out


def run(model_path, data_dir, tol=1e-3, eval_type="val", device="cuda:0"):
    import torch
    import yaml
    import b2t_ops as ops
    from rnn_trainer import BrainToTextDecoder_Trainer
    with open(os.path.join(model_path, "checkpoint", "args.yaml")) as f:
        args = yaml.safe_load(f)
    ck = torch.load(os.path.join(model_path, "checkpoint", "best_checkpoint"), weights_only=False, map_location="cpu")
    stored = float(ck["val_PER"]) if "val_PER" in ck else None
    tmp = tempfile.mkdtemp(prefix="accept_t15_")
    a = copy.deepcopy(args)
    a["mode"] = "eval"                       # no output directory, no training log file
    a["save_best_checkpoint"] = a["save_all_val_steps"] = a["save_final_model"] = False
    a["init_from_checkpoint"] = False
    a["output_dir"], a["checkpoint_dir"] = os.path.join(tmp, "out"), os.path.join(tmp, "out", "checkpoint")
    a["dataset"]["dataset_dir"] = data_dir
    a["num_training_batches"] = 1            # the training batch index is not used here; do not pre-generate 120,000 batches
    a["amd_bf16_matmul"] = False
    if eval_type != "val":
        raise ValueError("only the validation split carries the labels validation() needs")
    tr = BrainToTextDecoder_Trainer(a)
    missing, unexpected = tr.model.load_state_dict(strip_prefixes(ck["model_state_dict"]), strict=False)
    if missing or unexpected:
        raise RuntimeError(f"checkpoint does not fit the model: missing {missing}, unexpected {unexpected}")
    res = {}
    for name, amp in (("fp32", False), ("bf16", True)):
        ops.set_amp(amp)
        try:
            m = tr.validation(tr.val_loader)
        finally:
            ops.set_amp(False)
        res[name] = dict(PER=float(m["avg_PER"]), loss=float(m["avg_loss"]) if "avg_loss" in m else None)
    out = dict(checkpoint_val_PER=stored, **{f"PER_{k}": v["PER"] for k, v in res.items()}, tolerance_abs=tol,
               n_val_batches=len(tr.val_loader) if hasattr(tr.val_loader, "__len__") else None)
    ok = stored is not None and all(abs(v["PER"] - stored) <= tol for v in res.values())
    out["within_tolerance"] = bool(ok)
    return ok, out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--model_path", default=os.path.join(ROOT, "data", "t15_pretrained_rnn_baseline"))
    ap.add_argument("--data_dir", default=os.path.join(ROOT, "data", "hdf5_data_final"))
    ap.add_argument("--tol", type=float, default=1e-3)
    z = ap.parse_args()
    ok, why = available(z.model_path, z.data_dir)
    if not ok:
        print(json.dumps(dict(skipped=why)))
        sys.exit(3)
    ok, out = run(z.model_path, z.data_dir, z.tol)
    print(json.dumps(out))
    sys.exit(0 if ok else 1)
