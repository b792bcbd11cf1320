#!/usr/bin/env python3
"""Trajectory noise of the PER acceptance proxy (tests/test_gpu_trainer.py): the same fp32 training run under execution plans that
differ only in the ORDER of exact fp32 accumulations (fused / unfused projections, one / several time chunks), and the bf16 run."""
import os, sys, tempfile, importlib.util
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nejm-brain-to-text_amd")); sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("tt", os.path.join(ROOT, "tests", "test_gpu_trainer.py"))
TT = importlib.util.module_from_spec(spec); spec.loader.exec_module(TT)
import b2t_ops as ops
from rnn_trainer import BrainToTextDecoder_Trainer
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1600
variant = sys.argv[2] if len(sys.argv) > 2 else "f32"
amp = variant == "amp"
tr = BrainToTextDecoder_Trainer(TT._per_args(tempfile.mkdtemp(), amp, N))
st = tr.train()
print(f"{variant} env FUSED={os.environ.get('B2T_FUSED_PROJ')} CHUNKS={os.environ.get('B2T_CHUNKS')}: PER {st['val_PERs'][-1]:.5f} loss {st['val_losses'][-1]:.5f}")
