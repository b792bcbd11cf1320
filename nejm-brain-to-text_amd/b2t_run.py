"""b2t_run — run one of the reference's own launch scripts, UNCHANGED, on the MI355X path.

    python /path/to/nejm-brain-to-text_amd/b2t_run.py train_model.py
    python /path/to/nejm-brain-to-text_amd/b2t_run.py evaluate_model.py --model_path ... --data_dir ...
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        /path/to/nejm-brain-to-text_amd/b2t_run.py train_model.py

Why a launcher: `python train_model.py` puts the SCRIPT's directory at sys.path[0], ahead of PYTHONPATH, so the
reference's `from rnn_trainer import BrainToTextDecoder_Trainer` (model_training/train_model.py:2) and
`from rnn_model import GRUDecoder` / `from evaluate_model_helpers import *` (model_training/evaluate_model.py:12-13)
resolve to the reference's own files next to the script -- PYTHONPATH alone can never win that race.  This launcher
puts the package directory first, the script's directory right behind it (so that everything else the script imports
from its own directory still resolves), checks that every drop-in module name now resolves into the package, and runs
the script file as `__main__` with its own argv, from the caller's working directory (the scripts read `rnn_args.yaml`
and `../data/...` relative to it).  No reference file is copied, edited or imported by this module.
"""
from __future__ import annotations

import importlib.util
import os
import runpy
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))

# module names the reference's scripts import that this package replaces (model_training/*.py and the pybind module
# language_model/runtime/server/x86/python/lm_decoder.cc:51-75)
DROP_IN = ("rnn_model", "rnn_trainer", "data_augmentations", "dataset", "evaluate_model_helpers", "lm_decoder")


def resolve_check():
    """Every drop-in name must resolve to a file of this package (raises otherwise): a silent import of the reference's
    module would run the PyTorch path and report it as the HIP one."""
    for name in DROP_IN:
        loaded = sys.modules.get(name)
        origin = getattr(loaded, "__file__", None) if loaded is not None else None
        if origin is None:
            spec = importlib.util.find_spec(name)
            origin = spec.origin if spec is not None else None
        if origin is None or os.path.dirname(os.path.abspath(origin)) != PKG_DIR:
            raise ImportError(f"b2t_run: module '{name}' resolves to {origin!r}, not to {PKG_DIR}: the reference's own "
                              "module would be imported instead of the MI355X drop-in")


def run(script: str, argv):
    script = os.path.abspath(script)
    if not os.path.isfile(script):
        raise SystemExit(f"b2t_run: no such script: {script}")
    sdir = os.path.dirname(script)
    # drop what `python b2t_run.py` / `python -m b2t_run` put in front, then: package first, script directory second
    sys.path[:] = [p for p in sys.path if os.path.abspath(p or os.getcwd()) not in (PKG_DIR, sdir)]
    sys.path.insert(0, sdir)
    sys.path.insert(0, PKG_DIR)
    for name in DROP_IN:          # a module imported earlier from elsewhere must not shadow the drop-in
        m = sys.modules.get(name)
        if m is not None and os.path.dirname(os.path.abspath(getattr(m, "__file__", "") or "")) != PKG_DIR:
            del sys.modules[name]
    importlib.invalidate_caches()
    resolve_check()
    sys.argv = [script] + list(argv)
    runpy.run_path(script, run_name="__main__")
    resolve_check()               # and nothing the script did swapped them back


def main():
    if len(sys.argv) < 2 or sys.argv[1] in ("-h", "--help"):
        sys.stderr.write(__doc__)
        raise SystemExit(2)
    run(sys.argv[1], sys.argv[2:])


if __name__ == "__main__":
    main()
