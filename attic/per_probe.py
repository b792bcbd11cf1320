import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "nejm-brain-to-text_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import logging; logging.disable(logging.INFO)
import test_gpu_trainer as T
from rnn_trainer import BrainToTextDecoder_Trainer
import b2t_ops as ops
n = int(sys.argv[1]); lr = float(sys.argv[2])
for amp in (False, True):
    a = T._per_args(tempfile.mkdtemp(), amp, n)
    a['lr_max'] = a['lr_max_day'] = lr
    a['batches_per_val_step'] = max(1, n // 4)
    ops.set_amp(amp)
    t0 = time.time(); tr = BrainToTextDecoder_Trainer(a); st = tr.train()
    print("amp", amp, "PERs", [round(p, 4) for p in st['val_PERs']], "time", round(time.time() - t0, 1), flush=True)
