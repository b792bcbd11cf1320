"""The drop-in launch recipe: the reference's scripts import `rnn_trainer` / `rnn_model` /
`evaluate_model_helpers` by bare name (model_training/train_model.py:1-6, evaluate_model.py:12-13) and Python puts the
script's own directory first on sys.path, so `PYTHONPATH=<pkg> python train_model.py` silently imports the reference's
modules.  `b2t_run.py` must make the package's modules the ones imported.  CPU only: a
temporary directory holds a script of the same shape next to DECOY modules of the same names (no reference file is used)."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "nejm-brain-to-text_amd")

SCRIPT = textwrap.dedent("""
    import sys
    from rnn_trainer import BrainToTextDecoder_Trainer
    from rnn_model import GRUDecoder
    from evaluate_model_helpers import *
    import rnn_trainer, rnn_model, evaluate_model_helpers, sibling_helper
    print("ARGV", sys.argv[1:])
    print("NAME", __name__)
    for m in (rnn_trainer, rnn_model, evaluate_model_helpers, sibling_helper):
        print("MOD", m.__name__, m.__file__)
    print("TRAINER_CLASS", BrainToTextDecoder_Trainer.__module__, hasattr(BrainToTextDecoder_Trainer, "train"))
    print("HELPERS", "runSingleDecodingStep" in globals(), "rearrange_speech_logits_pt" in globals())
""")
DECOY = "DECOY = True\nclass BrainToTextDecoder_Trainer: pass\nclass GRUDecoder: pass\n"


def _make(tmp_path):
    (tmp_path / "train_model.py").write_text(SCRIPT)
    for name in ("rnn_trainer", "rnn_model", "evaluate_model_helpers", "data_augmentations", "dataset"):
        (tmp_path / f"{name}.py").write_text(DECOY)
    (tmp_path / "sibling_helper.py").write_text("X = 1\n")
    return str(tmp_path / "train_model.py")


def _mods(out):
    return {l.split()[1]: l.split()[2] for l in out.splitlines() if l.startswith("MOD ")}


def test_plain_pythonpath_imports_the_decoys(tmp_path):
    """The round-2 recipe, kept as the negative control: the script directory wins over PYTHONPATH."""
    script = _make(tmp_path)
    env = dict(os.environ, PYTHONPATH=PKG)
    r = subprocess.run([sys.executable, script], capture_output=True, text=True, cwd=str(tmp_path), env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    m = _mods(r.stdout)
    assert os.path.dirname(m["rnn_trainer"]) == str(tmp_path) and os.path.dirname(m["rnn_model"]) == str(tmp_path)


def test_launcher_imports_the_package_modules(tmp_path):
    script = _make(tmp_path)
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    for cmd in ([sys.executable, os.path.join(PKG, "b2t_run.py"), script, "--flag", "7"],
                [sys.executable, os.path.join(PKG, "b2t_run.py"), "train_model.py", "--flag", "7"]):   # relative, like the README
        r = subprocess.run(cmd, capture_output=True, text=True, cwd=str(tmp_path), env=env, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        m = _mods(r.stdout)
        for name in ("rnn_trainer", "rnn_model", "evaluate_model_helpers"):
            assert os.path.dirname(m[name]) == PKG, (name, m[name])
        assert os.path.dirname(m["sibling_helper"]) == str(tmp_path)        # the script's other local imports still resolve
        assert "ARGV ['--flag', '7']" in r.stdout and "NAME __main__" in r.stdout
        assert "TRAINER_CLASS rnn_trainer True" in r.stdout and "HELPERS True True" in r.stdout


def test_launcher_as_module_with_pythonpath(tmp_path):
    """`python -m b2t_run train_model.py` with the package on PYTHONPATH (the form torch.distributed.run users need)."""
    script = _make(tmp_path)
    env = dict(os.environ, PYTHONPATH=PKG)
    r = subprocess.run([sys.executable, "-m", "b2t_run", script], capture_output=True, text=True, cwd=str(tmp_path), env=env,
                       timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    m = _mods(r.stdout)
    assert os.path.dirname(m["rnn_trainer"]) == PKG and os.path.dirname(m["rnn_model"]) == PKG


def test_launcher_refuses_when_a_drop_in_is_shadowed(tmp_path):
    """If a drop-in name cannot resolve into the package the launcher raises instead of running the wrong code."""
    sys.path.insert(0, PKG)
    try:
        import b2t_run
    finally:
        sys.path.pop(0)
    saved = b2t_run.DROP_IN
    b2t_run.DROP_IN = saved + ("no_such_drop_in_module",)
    try:
        try:
            b2t_run.resolve_check()
            raised = False
        except ImportError:
            raised = True
    finally:
        b2t_run.DROP_IN = saved
    assert raised
