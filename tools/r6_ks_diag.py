"""Where do non-finite values appear in the K-split wavefront?  Runs a shape repeatedly and reports (tensor, layer, first / last bad step)."""
import os, sys, json, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nejm-brain-to-text_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_gpu_wave as tw
import b2t_native as N, b2t_ops as ops
lib, P = N.load(), ops._p
for (L, T, B, H, p) in ((5, 10, 64, 512, 0.0), (5, 25, 64, 512, 0.4), (5, 300, 64, 512, 0.0)):
    bad = 0
    _, got = tw._run(L, T, B, H, p, seed=7, reference=False)
    d, keep = got["desc"], got["keep"]
    wsf, wsb, err = keep[-3], keep[-2], keep[-1]
    for rep in range(12):
        if rep:
            for t in got["out"] + got["dG"] + got["res"]: t.fill_(float("nan"))
            N.check(lib.b2t_gru_wave_fwd_f32(C.byref(d), P(wsf), P(err), ops._stream()), "fwd")
            N.check(lib.b2t_gru_wave_bwd_f32(C.byref(d), P(wsb), P(err), ops._stream()), "bwd")
            torch.cuda.synchronize()
        rep_bad = []
        for name in ("out", "res", "dG"):
            for l in range(L):
                x = got[name][l]
                nf = ~torch.isfinite(x).reshape(T, -1).all(dim=1)
                if bool(nf.any()):
                    idx = torch.nonzero(nf).flatten()
                    rows = ~torch.isfinite(x[int(idx[0])]).reshape(B, -1).all(dim=1)
                    rep_bad.append((name, l, int(idx[0]), int(idx[-1]), int(nf.sum()), int(rows.sum())))
        if not bool(torch.isfinite(got["dh_init"]).all()): rep_bad.append(("dh_init",))
        if rep_bad: bad += 1; print("KSDIAG", os.environ.get("B2T_LIB", "default").split("/")[-1], (L, T, B, H, p), "rep", rep, "err", int(err[0]), rep_bad[:6], flush=True)
    print("KSDIAG", os.environ.get("B2T_LIB", "default").split("/")[-1], (L, T, B, H, p), "bad reps", bad, "of 12", "dbg", [hex(int(v) & 0xffffffff) for v in err[8:13]], flush=True)
