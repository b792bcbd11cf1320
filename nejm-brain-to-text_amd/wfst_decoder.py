"""Host side of the WFST decoder: drives the batched token-passing kernels (csrc/wfst.hip) for U utterances and turns
what they leave in HBM into the reference's outputs.

    WfstSearch.search(logp)      CtcWfstBeamSearch::Search        ctc_wfst_beam_search.cc:70-121  (partial best path)
    WfstSearch.finalize()        CtcWfstBeamSearch::FinalizeSearch ctc_wfst_beam_search.cc:123-160 (n-best)

The per-frame work (blank skipping, ProcessEmitting / ProcessNonemitting, cut-offs, token hash) and the backward lattice
pruning run on the GPU; the n-best extraction from the pruned lattice (a few thousand arcs per utterance, once per
utterance) is the C++ host function b2t_lattice_nbest_host.
"""
from __future__ import annotations

import ctypes as C
from typing import List

import numpy as np
import torch

import b2t_native as N
import b2t_ops as ops


def _pow2_at_least(n: int) -> int:
    p = 1
    while p < n:
        p *= 2
    return p


class WfstSearch:
    def __init__(self, graph, opts, U: int = 1, device="cuda:0", max_frames: int = 1024, max_tokens: int = 1 << 19,
                 max_links: int = 1 << 21, hash_size: int = 0):
        """graph: wfst.DecodeGraph.  opts: an object with the reference's DecodeOptions fields (max_active, min_active, beam,
        lattice_beam, acoustic_scale, ctc_blank_skip_threshold, length_penalty, nbest)."""
        self.g, self.U, self.device = graph, int(U), torch.device(device)
        self.lib = N.load()
        d = graph.to_device(self.device)
        self._keep = d
        self.cg = N.WfstGraph(d["row"].data_ptr(), d["ilabel"].data_ptr(), d["olabel"].data_ptr(), d["weight"].data_ptr(),
                              d["next"].data_ptr(), d["n_eps"].data_ptr(), d["final"].data_ptr(), graph.n_states, graph.start)
        if hash_size <= 0:   # a frame can hold at most one token per graph state
            hash_size = max(1024, min(1 << 16, _pow2_at_least(2 * min(graph.n_states, 1 << 15))))
        self.caps = (int(max_frames), int(max_tokens), int(max_links), int(hash_size))
        self.set_opts(opts)
        self.state_bytes = self.lib.b2t_wfst_state_bytes(*self.caps)
        self.state = torch.zeros((self.U * self.state_bytes,), dtype=torch.uint8, device=self.device)
        off = (N.LL * 16)()
        N.check(self.lib.b2t_wfst_state_offsets(*self.caps, off), "b2t_wfst_state_offsets")
        self.off = list(off)
        self.reset()

    def set_opts(self, opts):
        self.nbest = int(getattr(opts, "nbest", 10))
        self.acoustic_scale = float(opts.acoustic_scale)
        self.lattice_beam = float(opts.lattice_beam)
        mf, mt, ml, hs = self.caps
        self.co = N.WfstOpts(float(opts.beam), float(opts.lattice_beam), float(getattr(opts, "beam_delta", 0.5)), float(opts.acoustic_scale),
                             float(getattr(opts, "length_penalty", 0.0)), float(opts.ctc_blank_skip_threshold),
                             int(min(opts.max_active, 2 ** 31 - 1)), int(opts.min_active), mf, mt, ml, hs)

    def _s(self):
        return ops._stream()

    def reset(self):
        with torch.cuda.device(self.device):
            N.check(self.lib.b2t_wfst_reset(C.byref(self.cg), C.byref(self.co), ops._p(self.state), self.U, self._s()), "b2t_wfst_reset")
        self.finalized = False

    # ---- Search ------------------------------------------------------------------------------------------------------
    def search(self, logp: torch.Tensor, lens=None):
        """logp [U, T, C] fp32 on the device (after the DecodeNumpy prologue); may be called chunk after chunk."""
        ops._need(logp, name="logp")
        U, T, Cc = logp.shape
        if U != self.U:
            raise ValueError(f"expected {self.U} utterances")
        lens_t = None if lens is None else torch.as_tensor(lens, dtype=torch.int32).to(self.device).contiguous()
        with torch.cuda.device(self.device):
            N.check(self.lib.b2t_wfst_search_f32(C.byref(self.cg), C.byref(self.co), ops._p(self.state), ops._p(logp),
                                                 ops._p(lens_t), U, T, Cc, self._s()), "b2t_wfst_search_f32")

    def best_path(self, use_final: bool = False, max_len: int = 0):
        """[(inputs, times, words, lm_score, ac_score)] per utterance: GetBestPath + ConvertToInputs + the likelihood pair."""
        max_len = max_len or self.caps[0] * 2
        U, dev = self.U, self.device
        ali = torch.zeros((U, max_len), dtype=torch.int32, device=dev); fr = torch.zeros_like(ali); wd = torch.zeros_like(ali)
        na = torch.zeros((U,), dtype=torch.int32, device=dev); nw = torch.zeros_like(na)
        cs = torch.zeros((U, 2), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            N.check(self.lib.b2t_wfst_best_path(C.byref(self.cg), C.byref(self.co), ops._p(self.state), U, int(use_final), max_len,
                                                ops._p(ali), ops._p(fr), ops._p(na), ops._p(wd), ops._p(nw), ops._p(cs), self._s()),
                    "b2t_wfst_best_path")
        self._check_overflow()
        ali, fr, wd, na, nw, cs = (t.cpu().numpy() for t in (ali, fr, wd, na, nw, cs))
        out = []
        for u in range(U):
            inp, tm = convert_to_inputs(ali[u, :na[u]], fr[u, :na[u]])
            out.append((inp, tm, [int(w) for w in wd[u, :nw[u]]], -float(cs[u, 0]), -float(cs[u, 1])))
        return out

    def _header(self):
        st = self.state.view(self.U, self.state_bytes)
        return st[:, self.off[0]:self.off[0] + 40].contiguous().view(torch.int32).cpu().numpy()

    def _check_overflow(self):
        h = self._header()
        if h[:, 3].any():
            bits = int(np.bitwise_or.reduce(h[:, 3]))
            what = [n for b, n in ((1, "max_tokens"), (2, "max_links"), (4, "hash_size"), (8, "max_frames")) if bits & b]
            raise RuntimeError(f"WFST search: a capacity was exhausted ({', '.join(what)} of {self.caps}; peak tokens "
                               f"{int(h[:, 1].max())}, links {int(h[:, 2].max())}); results are invalid -- construct WfstSearch "
                               "with larger capacities")

    def frames_decoded(self):
        return [int(v) for v in self._header()[:, 0]]

    # ---- FinalizeSearch ------------------------------------------------------------------------------------------------
    def finalize(self):
        """FinalizeDecoding on the GPU, then per utterance the n-best word sequences of the pruned lattice:
        [[(inputs, times, words, lm_score, ac_score), ...] best first]."""
        with torch.cuda.device(self.device):
            N.check(self.lib.b2t_wfst_finalize(C.byref(self.cg), C.byref(self.co), ops._p(self.state), self.U, self._s()), "b2t_wfst_finalize")
        self.finalized = True
        self._check_overflow()
        if self.nbest == 1:
            return [[r] if self.frames_decoded()[u] > 0 else [] for u, r in enumerate(self.best_path(True))]
        hdr = self._header()
        out = []
        for u in range(self.U):
            out.append(self._nbest_of(u, hdr[u]))
        return out

    def _arr(self, u, which, count, dtype):
        base = u * self.state_bytes + self.off[which]
        nb = count * np.dtype(dtype).itemsize
        return np.frombuffer(self.state[base:base + nb].cpu().numpy().tobytes(), dtype=dtype)

    def _nbest_of(self, u, h):
        F, n_tok, n_link = int(h[0]), int(h[1]), int(h[2])
        if F == 0:
            return []
        mf = self.caps[0]
        tok_off = self._arr(u, 2, F + 2, np.int32); link_off = self._arr(u, 3, 2 * F + 2, np.int32)
        mapping = self._arr(u, 1, F, np.int32); cost_off = self._arr(u, 4, F, np.float32)
        tok_state = self._arr(u, 5, n_tok, np.int32); tok_extra = self._arr(u, 7, n_tok, np.float32)
        src, dst, arc = (self._arr(u, k, n_link, np.int32) for k in (8, 9, 10))
        ac, gr = self._arr(u, 11, n_link, np.float32), self._arr(u, 12, n_link, np.float32)
        alive = self._arr(u, 13, n_link, np.uint8).astype(bool)
        # acoustic cost of an emitting link has the frame's cost offset in it (GetRawLattice takes it out, :150-160)
        il, ol = self.g.ilabel[arc], self.g.olabel[arc]
        frame_of_link = np.zeros(n_link, dtype=np.int64)
        for f in range(F):
            frame_of_link[link_off[2 * f + 1]:link_off[2 * f + 2]] = f
        ac = np.where(il != 0, ac - cost_off[frame_of_link], ac).astype(np.float32)
        keep = alive & np.isfinite(tok_extra[src]) & np.isfinite(tok_extra[dst])
        t0, t1 = int(tok_off[F]), int(tok_off[F + 1])
        has_final = bool(h[9])
        last = np.arange(t0, t1, dtype=np.int32)
        fc = self.g.final[tok_state[t0:t1]] if has_final else np.zeros(t1 - t0, dtype=np.float32)
        ok = np.isfinite(fc) & np.isfinite(tok_extra[t0:t1])
        fs, fcost = np.ascontiguousarray(last[ok]), np.ascontiguousarray(fc[ok].astype(np.float32))
        a = [np.ascontiguousarray(x[keep]) for x in (src, dst, il.astype(np.int32), ol.astype(np.int32), gr, ac)]
        nb = self.nbest
        w_cap = a_cap = nb * (2 * F + 16) + 16
        ow = np.zeros(w_cap, dtype=np.int32); oa = np.zeros(a_cap, dtype=np.int32)
        woff = np.zeros(nb + 1, dtype=np.int32); aoff = np.zeros(nb + 1, dtype=np.int32); costs = np.zeros(2 * nb, dtype=np.float32)
        P = lambda x: x.ctypes.data_as(C.c_void_p)
        n = self.lib.b2t_lattice_nbest_host(n_tok, 0, int(a[0].shape[0]), P(a[0]), P(a[1]), P(a[2]), P(a[3]), P(a[4]), P(a[5]),
                                            int(fs.shape[0]), P(fs), P(fcost), nb, C.c_float(self.lattice_beam), P(ow), P(woff), w_cap,
                                            P(oa), P(aoff), a_cap, P(costs))
        if n < 0:
            raise RuntimeError("b2t_lattice_nbest_host failed: " + N.last_error())
        res = []
        for k in range(n):
            ali = oa[aoff[k]:aoff[k + 1]]
            inp, tm = convert_to_inputs(ali, mapping[:len(ali)] if len(ali) == F else np.arange(len(ali)))
            res.append((inp, tm, [int(w) for w in ow[woff[k]:woff[k + 1]]], -float(costs[2 * k]), -float(costs[2 * k + 1])))
        return res


def convert_to_inputs(alignment, frames):
    """CtcWfstBeamSearch::ConvertToInputs (ctc_wfst_beam_search.cc:162-188): drop blanks (ilabel 1), merge repeats, ilabel - 1;
    the time of a unit is the input frame of its LAST repeated label."""
    inp: List[int] = []; tm: List[int] = []
    cur, n = 0, len(alignment)
    while cur < n:
        while cur < n and alignment[cur] - 1 == 0:
            cur += 1
        while cur + 1 < n and alignment[cur + 1] == alignment[cur]:
            cur += 1
        if cur < n:
            inp.append(int(alignment[cur]) - 1); tm.append(int(frames[cur])); cur += 1
    return inp, tm
