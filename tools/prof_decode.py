#!/usr/bin/env python3
"""The decode kernels under rocprofv3 (tools/run_prof_r3.sh): 3 repetitions of the tools/bench_wfst.py offline workload
(32 utterances, cluster search with PruneActiveTokens every 25 frames, finalize + lattice) and 3 of the lexicon prefix
beam 10/100 over the same log-probabilities.  Prints the algorithmic bytes per search launch for the roofline lines."""
import json
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "nejm-brain-to-text_amd"), ROOT, os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import b2t_native as N      # noqa: E402
import b2t_ops as ops       # noqa: E402
import bench_wfst as BW     # noqa: E402
import bench_secondary as BS   # noqa: E402
import ngram_lm             # noqa: E402
from wfst_decoder import WfstSearch   # noqa: E402

lib = N.load(); dev = torch.device("cuda:0")
prons, words, arpa, g, seqs, logits, lens, _ = BW.make()
U, T, C = logits.shape
_, _, lp = BW._logp(logits, dev, lib)
S = WfstSearch(g, BW.Opt, U=U, max_frames=T + 8, max_tokens=1 << 22, max_links=1 << 24, prune_interval=25, prune_min_fill=0.0)
for rep in range(3):
    S.reset(); S.search(lp, lens)
    mem = S.memory_stats(); arcs = S.arcs_expanded()
    S.finalize()
created_tok = sum(m["created_tokens"] for m in mem); created_link = sum(m["created_links"] for m in mem)
alg = 16.0 * sum(arcs) + 20.0 * created_tok + 21.0 * created_link
print(json.dumps(dict(search_launches_per_rep=5, utterances=U, frames=int(T), algorithmic_bytes_per_rep=alg,
                      algorithmic_bytes_per_search_launch=alg / 5, graph_bytes=g.nbytes())))
# lexicon prefix beam 10 / 100, 32 utterances per call
import ctypes as Ct
lex = ngram_lm.Lexicon(prons, C); wlm = ngram_lm.SparseNGramLM.from_arpa(arpa, lex.words)
dl, dm = lex.to_device(dev), wlm.to_device(dev)
d = N.LexLmDesc(dl["child"].data_ptr(), dl["wbeg"].data_ptr(), dl["wend"].data_ptr(), dl["wlist"].data_ptr(),
                dm["cb"].data_ptr(), dm["ce"].data_ptr(), dm["ctok"].data_ptr(), dm["cnode"].data_ptr(),
                dm["logp"].data_ptr(), dm["bow"].data_ptr(), dm["suffix"].data_ptr(), dm["nstate"].data_ptr(),
                wlm.start_state, wlm.eos, 1, float(1.0 / 0.325), 0.0, float(wlm.unk_logp))
b = BS._beam_buffers(lib, U, T, 100, dev)
lens_t = torch.from_numpy(lens.astype(np.int32)).to(dev)
_p = ops._p
for rep in range(3):
    N.check(lib.b2t_beam_reset(_p(b["state"]), U, b["L"], b["NN"], ops._stream()), "reset")
    N.check(lib.b2t_prefix_beam_search_lex_f32(_p(lp), _p(lens_t), U, T, C, 10, 100, 0, _p(b["state"]), b["L"], b["NN"], _p(b["hyps"]),
                                               _p(b["hl"]), _p(b["sc"]), _p(b["vs"]), _p(b["tm"]), Ct.byref(d), _p(b["lms"]), ops._stream()), "lex")
torch.cuda.synchronize()
print("done")
