"""Save the pruned token lattices of a few utterances of the tools/bench_wfst.py workload (for profiling csrc/lattice.cpp on the host)."""
import math, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "nejm-brain-to-text_amd")); sys.path.insert(0, ROOT)
import bench_wfst as B
import b2t_native as N, b2t_ops as ops
from wfst_decoder import WfstSearch
lib = N.load(); dev = torch.device("cuda:0")
prons, words, arpa, g, seqs, logits, lens, build_s = B.make()
U, T, C = logits.shape
lg = torch.from_numpy(logits).to(dev); pri = torch.zeros_like(lg); lp = torch.empty_like(lg)
N.check(lib.b2t_lm_prologue_f32(ops._p(lg), ops._p(pri), float(math.log(90.0)), ops._p(lp), U * T, C, ops._stream()), "prologue")
S = WfstSearch(g, B.Opt, U=U, max_frames=T + 8, max_tokens=1 << 22, max_links=1 << 24)
S.reset(); S.search(lp, lens)
S.finalize_gpu() if hasattr(S, "finalize_gpu") else None
fin = S.finalize()
hdr = S._header()
cn, ((src, dst, il, ol, gr, ac, fs, fc), a_off, f_off) = S._lattices()   # flat arrays, utterance u at [a_off[u], a_off[u + 1])
out = {}
for u in range(int(os.environ.get("B2T_DUMP_U", "4"))):
    n_states, n_arcs, n_final, start = (int(v) for v in cn[u, :4])
    out.update({f"u{u}_{k}": v for k, v in dict(src=src[a_off[u]:a_off[u + 1]], dst=dst[a_off[u]:a_off[u + 1]], il=il[a_off[u]:a_off[u + 1]], ol=ol[a_off[u]:a_off[u + 1]], gr=gr[a_off[u]:a_off[u + 1]],
                                                 ac=ac[a_off[u]:a_off[u + 1]], fs=fs[f_off[u]:f_off[u + 1]], fc=fc[f_off[u]:f_off[u + 1]], meta=np.array([n_states, n_arcs, n_final, start, int(hdr[u, 0])])).items()})
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "lattices.npz"), **out)
print("saved", {k: v.shape for k, v in out.items() if k.endswith("meta")}, "...")
