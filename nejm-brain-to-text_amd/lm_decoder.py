"""lm_decoder — Python surface of the reference's pybind11 module
(language_model/runtime/server/x86/python/lm_decoder.cc:51-75) over the HIP decoder kernels.

Two searchers, chosen like the reference's BrainSpeechDecoder does (brain_speech_decoder.cc:23-28):
  * a decode graph is loaded (DecodeResource(fst_path, ..) -- an OpenFST vector/standard TLG.fst or an .npz written by
    wfst.save_graph -- or DecodeResource.set_graph(wfst.build_tlg(lexicon, arpa))): CtcWfstBeamSearch, i.e. Kaldi's
    lattice-generating token passing over T o L o G, as batched GPU kernels (csrc/wfst.hip, wfst_decoder.WfstSearch):
    partial best path after every Decode(), n-best word sequences with separate graph / acoustic scores after
    FinishDecoding(), Rescore() with a second grammar;
  * no graph: the LM-free CtcPrefixBeamSearch (csrc/beam.hip), optionally fused with a token-level ARPA n-gram
    (DecodeResource.set_token_lm) or constrained by a lexicon + word n-gram (set_lexicon_lm).
The DecodeNumpy prologue (log_softmax - priors, blank penalty) is a kernel of its own.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional

import numpy as np
import torch

import b2t_native as N
import b2t_ops as ops

K_SPACE = "▁"   # WeNet's kSpaceSymbol


class DecodeOptions:
    """DecodeOptions(max_active, min_active, beam, lattice_beam, acoustic_scale, ctc_blank_skip_threshold,
    length_penalty, nbest) — same 8 positional arguments as the reference (language-model-standalone.py:486-496).
    The WFST fields are stored; the prefix-beam searcher uses first_beam_size / second_beam_size (defaults 10/10)."""

    def __init__(self, max_active, min_active, beam, lattice_beam, acoustic_scale, ctc_blank_skip_threshold,
                 length_penalty, nbest):
        self.max_active, self.min_active, self.beam, self.lattice_beam = max_active, min_active, beam, lattice_beam
        self.acoustic_scale, self.ctc_blank_skip_threshold = acoustic_scale, ctc_blank_skip_threshold
        self.length_penalty, self.nbest = length_penalty, nbest
        self.first_beam_size, self.second_beam_size, self.blank = 10, 10, 0
        self.lm_alpha, self.lm_beta, self.lm_eos = 0.0, 0.0, False   # token-level n-gram fusion (set_token_lm)


class DecodeResource:
    """DecodeResource(fst_path, lm_fst_path, rescore_lm_fst_path, dict_path, unit_path)."""

    def __init__(self, fst_path, lm_fst_path, rescore_lm_fst_path, dict_path, unit_path):
        import wfst
        self.graph = wfst.graph_from_files(fst_path, dict_path) if fst_path else None
        # the grammars used by Rescore(): G.fst (its scores are taken out) and G_no_prune.fst (its scores are put in)
        # (read and kept as CSR arrays in C++, arc-sorted once: G_no_prune.fst has 10^7-10^9 arcs in the reference's setup)
        self.symbols = self._read_table(dict_path) if dict_path else None
        # ReadAndPrepareLmFst (kaldi-fst-io.cc:129-147, called at brain_speech_decoder.h:59,71): a grammar with #0 on the input
        # side of its back-off arcs (eps2disambig.pl, make_tlg.sh:35-38) is projected on its output labels, so the back-off
        # label is epsilon afterwards; an acceptor (already projected, or compiled without that step) is used as it is.  The
        # label is determined from the arcs, not from words.txt listing "#0" (Kaldi's word tables almost always do).
        ids = [i for i, w in (self.symbols or {}).items() if w == "#0"]
        disambig = ids[0] if ids else None
        self.lm_fst = self.rescore_lm_fst = None
        self.backoff_label = None
        labels = []
        if lm_fst_path:
            self.lm_fst, b = wfst.HostFst.read_openfst(lm_fst_path).prepare_lm(disambig); labels.append(b)
        if rescore_lm_fst_path:
            self.rescore_lm_fst, b = wfst.HostFst.read_openfst(rescore_lm_fst_path).prepare_lm(disambig); labels.append(b)
        if self.lm_fst is not None and self.rescore_lm_fst is not None:
            if labels[0] != labels[1]:
                raise ValueError(f"G.fst backs off through label {labels[0]} and the rescoring grammar through {labels[1]}: "
                                 "compile both the same way (with or without eps2disambig)")
            self.backoff_label = labels[0]
        self.units = self._read_table(unit_path) if unit_path else None
        self.token_lm = None
        self.lexicon = self.word_lm = None
        self.sil = 1

    def set_graph(self, graph):
        """Attach a wfst.DecodeGraph built in memory (wfst.build_tlg); its word table becomes the symbol table."""
        self.graph = graph
        self.symbols = {i: w for i, w in enumerate(graph.words)}
        if "#0" in graph.words:
            self.backoff_label = graph.words.index("#0")

    def set_rescore_grammars(self, lm_fst, rescore_lm_fst, backoff_label):
        """Grammars for Rescore() (wfst.Fst or wfst.HostFst): the one composed into the graph and the one to rescore with."""
        import wfst
        conv = lambda g: g if isinstance(g, wfst.HostFst) else wfst.HostFst.from_fst(g).arcsort()
        self.lm_fst, self.rescore_lm_fst, self.backoff_label = conv(lm_fst), conv(rescore_lm_fst), backoff_label

    def set_token_lm(self, lm):
        """Attach an ngram_lm.NGramLM over the decoder's output tokens (fused into the prefix beam search)."""
        self.token_lm = lm

    def set_lexicon_lm(self, lexicon, word_lm, sil: int = 1):
        """Attach an ngram_lm.Lexicon and an ngram_lm.SparseNGramLM over its words: the search is then constrained to
        SIL-delimited dictionary words and scored by the word n-gram (b2t_prefix_beam_search_lex_f32)."""
        if word_lm.W != len(lexicon.words):
            raise ValueError("the word LM was built for a different vocabulary than the lexicon")
        self.lexicon, self.word_lm, self.sil = lexicon, word_lm, sil

    @staticmethod
    def _read_table(path):
        table = {}
        with open(path) as f:
            for line in f:
                parts = line.split()
                if len(parts) >= 2:
                    table[int(parts[-1])] = parts[0]
        return table


class DecodeResult:
    def __init__(self, sentence="", ac_score=0.0, lm_score=0.0):
        self.sentence, self.ac_score, self.lm_score = sentence, ac_score, lm_score


def process_blank(s: str) -> str:
    """language_model/runtime/core/utils/string.cc:121-146: drop leading/duplicate/trailing space symbols, lower-case."""
    out = []
    for ch in s:
        if ch != K_SPACE and ch != " ":
            out.append(ch)
        elif out and out[-1] != " ":
            out.append(" ")
    return "".join(out).rstrip(" ").lower()


class BrainSpeechDecoder:
    """Decode()/Reset()/FinishDecoding()/DecodedSomething()/result() as in brain_speech_decoder.h:112-124.
    Decode() may be called repeatedly with consecutive chunks of log-probabilities (streaming)."""

    def __init__(self, resource: DecodeResource, opts: DecodeOptions, device="cuda:0", max_len=1024):
        self.res, self.opts = resource, opts
        self.device = torch.device(device)
        self.max_len = max_len
        self.wfst = None
        if resource.graph is not None:      # CtcWfstBeamSearch (brain_speech_decoder.cc:26-28)
            from wfst_decoder import WfstSearch
            # (a PruneActiveTokens pass that falls due runs behind the chunk's partial result, not inside the chunk: same lattice)
            self.wfst = WfstSearch(resource.graph, opts, U=1, device=self.device, max_frames=max_len, prune_after_read=True)
            self.acoustic_scale = float(opts.acoustic_scale)
            self._result = []
            self._entries = []
            return
        self.max_nodes = max_len * max(1, opts.second_beam_size) + 2
        lib = N.load()
        nbytes = lib.b2t_beam_state_bytes(self.max_len, self.max_nodes)
        self.state = torch.empty((nbytes,), dtype=torch.uint8, device=self.device)
        self.acoustic_scale = 1.0
        self._result: List[DecodeResult] = []
        self._out = None
        self.Reset()

    def SetOpt(self, opts: DecodeOptions):
        self.opts = opts
        if self.wfst is not None:
            self.wfst.set_opts(opts)
            self.acoustic_scale = float(opts.acoustic_scale)

    def _wfst_results(self, entries):
        """UpdateResult (brain_speech_decoder.cc:113-137): word strings, ac_score = acoustic / acoustic_scale, lm_score = graph."""
        table = self.res.symbols or {}
        self._entries = entries
        self._result = []
        for inp, tm, words, lm, ac in entries:
            r = DecodeResult(process_blank("".join(" " + table.get(int(w), str(int(w))) for w in words)), ac / self.acoustic_scale, lm)
            r.tokens, r.times, r.word_ids = np.asarray(inp), np.asarray(tm), list(words)
            self._result.append(r)

    def Reset(self):
        self._result = []
        if self.wfst is not None:
            self.wfst.reset()
            return
        with torch.cuda.device(self.device):
            N.check(N.load().b2t_beam_reset(ops._p(self.state), 1, self.max_len, self.max_nodes, ops._stream()),
                    "b2t_beam_reset")

    def Decode(self, logp):
        lp = torch.as_tensor(logp, dtype=torch.float32).to(self.device).contiguous()
        if lp.dim() != 2:
            raise ValueError("logp must be [T, C]")
        if self.wfst is not None:
            if lp.shape[0] == 0:
                return
            self.wfst.search(lp.unsqueeze(0))
            bp = self.wfst.best_path(False)[0]
            self._wfst_results([bp] if self.wfst.frames_decoded()[0] > 0 else [])
            return
        T, Cc = lp.shape
        bm = self.opts.second_beam_size
        hyps = torch.zeros((1, bm, self.max_len), dtype=torch.int32, device=self.device)
        hl = torch.empty((1, bm), dtype=torch.int32, device=self.device)
        sc = torch.empty((1, bm), dtype=torch.float32, device=self.device)
        vs = torch.empty((1, bm), dtype=torch.float32, device=self.device)
        tm = torch.zeros((1, bm, self.max_len), dtype=torch.int32, device=self.device)
        lm = self.res.token_lm
        lms = torch.zeros((1, bm), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            if self.res.lexicon is not None:
                lex, wlm = self.res.lexicon, self.res.word_lm
                if lex.C != Cc:
                    raise ValueError(f"the lexicon was built for {lex.C} classes, the log-probabilities have {Cc}")
                dl, dm = lex.to_device(self.device), wlm.to_device(self.device)
                d = N.LexLmDesc(dl["child"].data_ptr(), dl["wbeg"].data_ptr(), dl["wend"].data_ptr(), dl["wlist"].data_ptr(),
                                dm["cb"].data_ptr(), dm["ce"].data_ptr(), dm["ctok"].data_ptr(), dm["cnode"].data_ptr(),
                                dm["logp"].data_ptr(), dm["bow"].data_ptr(), dm["suffix"].data_ptr(), dm["nstate"].data_ptr(),
                                wlm.start_state, wlm.eos if self.opts.lm_eos else -1, self.res.sil, float(self.opts.lm_alpha),
                                float(self.opts.lm_beta), float(wlm.unk_logp))
                N.check(N.load().b2t_prefix_beam_search_lex_f32(
                    ops._p(lp), None, 1, T, Cc, self.opts.first_beam_size, bm, self.opts.blank, ops._p(self.state),
                    self.max_len, self.max_nodes, ops._p(hyps), ops._p(hl), ops._p(sc), ops._p(vs), ops._p(tm), C.byref(d),
                    ops._p(lms), ops._stream()), "b2t_prefix_beam_search_lex_f32")
            elif lm is None:
                N.check(N.load().b2t_prefix_beam_search_f32(ops._p(lp), None, 1, T, Cc, self.opts.first_beam_size, bm,
                                                            self.opts.blank, ops._p(self.state), self.max_len,
                                                            self.max_nodes, ops._p(hyps), ops._p(hl), ops._p(sc), ops._p(vs),
                                                            ops._p(tm), ops._stream()), "b2t_prefix_beam_search_f32")
            else:
                if lm.C != Cc:
                    raise ValueError(f"the token LM was built for {lm.C} classes, the log-probabilities have {Cc}")
                d = lm.to_device(self.device)
                N.check(N.load().b2t_prefix_beam_search_lm_f32(
                    ops._p(lp), None, 1, T, Cc, self.opts.first_beam_size, bm, self.opts.blank, ops._p(self.state),
                    self.max_len, self.max_nodes, ops._p(hyps), ops._p(hl), ops._p(sc), ops._p(vs), ops._p(tm),
                    ops._p(d["child"]), ops._p(d["logp"]), ops._p(d["bow"]), ops._p(d["suffix"]), ops._p(d["nstate"]), lm.V,
                    lm.start_state, lm.eos if self.opts.lm_eos else -1, float(self.opts.lm_alpha), float(self.opts.lm_beta),
                    float(lm.unk_logp), ops._p(lms), ops._stream()), "b2t_prefix_beam_search_lm_f32")
        self._out = (hyps.cpu().numpy()[0], hl.cpu().numpy()[0], sc.cpu().numpy()[0], vs.cpu().numpy()[0],
                     tm.cpu().numpy()[0], lms.cpu().numpy()[0])
        self._update_result()

    def _update_result(self):
        hyps, hl, sc, vs, tm, lms = self._out
        table = self.res.symbols or self.res.units
        fused = self.res.token_lm is not None or self.res.lexicon is not None
        self._result = []
        for i in range(len(hl)):
            if hl[i] < 0:
                continue
            ids = hyps[i, :hl[i]]
            if self.res.lexicon is not None:
                if not np.isfinite(lms[i]):
                    continue                     # ends inside a word
                import ngram_lm
                words, _ = ngram_lm.replay_words(self.res.lexicon, self.res.word_lm, ids, self.opts.lm_alpha,
                                                 self.opts.lm_beta, self.res.sil, self.opts.lm_eos)
                if words is None:
                    continue
            else:
                words = [table.get(int(t), str(int(t))) if table else str(int(t)) for t in ids]
            r = DecodeResult(process_blank("".join(" " + w for w in words)), float(sc[i]) / self.acoustic_scale,
                             float(lms[i]) if fused else float(sc[i]))
            r.tokens, r.times, r.viterbi_score = ids.copy(), tm[i, :hl[i]].copy(), float(vs[i])
            r.total_score = float(sc[i]) + (float(lms[i]) if fused else 0.0)
            self._result.append(r)
        if fused:   # the end-of-sentence term (lm_eos) can reorder the final list
            self._result.sort(key=lambda r: -r.total_score)

    def FinishDecoding(self):
        if self.wfst is not None:           # FinalizeSearch + UpdateResult (brain_speech_decoder.cc:41-44)
            self._wfst_results(self.wfst.finalize()[0])
        # (CtcPrefixBeamSearch::FinalizeSearch is a no-op, ctc_prefix_beam_search.h)

    def Rescore(self, deep_list: int = 0):
        """brain_speech_decoder.cc:61-101: take the scores of the grammar that is composed into the graph out of the
        lattice and put the scores of the rescoring grammar in (LatticeRescore at scale -1, then at scale +1: two lattice
        compositions + determinisations), then list the n-best again.  Done as the reference does it, on the LATTICE
        (b2t_lattice_rescore_nbest_host, csrc/graphc.cpp): both grammars are determinised on the fly along the lattice's words,
        the lattice is composed with them, and the n-best distinct word sequences of the product are ranked by the new cost --
        every word sequence of the pruned lattice takes part, so the new grammar can promote one from arbitrarily far down.
        deep_list > 0 selects the round-2 approximation instead (the exchange on an n-best list of that length; kept for the
        comparison in tests/test_gpu_wfst.py)."""
        if self.wfst is None:
            raise RuntimeError("Rescore() needs the WFST searcher (load a decode graph)")
        if self.res.lm_fst is None or self.res.rescore_lm_fst is None or self.res.backoff_label is None:
            raise RuntimeError("Rescore() needs both grammars (DecodeResource lm_fst_path / rescore_lm_fst_path or set_rescore_grammars)")
        keep = len(self._result)
        if not deep_list:
            self._wfst_results(self.wfst._nbest_all(keep, rescore=(self.res.lm_fst, self.res.rescore_lm_fst, self.res.backoff_label))[0])
            return
        deep = self.wfst._nbest_all(max(deep_list, keep))[0]
        rescored = []
        for inp, tm, words, lm, ac in deep:
            g_old = self.res.lm_fst.grammar_score(words, self.res.backoff_label)
            g_new = self.res.rescore_lm_fst.grammar_score(words, self.res.backoff_label)
            if not np.isfinite(g_new):
                continue
            rescored.append((inp, tm, words, lm + g_old - g_new, ac))
        rescored.sort(key=lambda e: -(e[3] + e[4]))
        self._wfst_results(rescored[:keep])

    def DecodedSomething(self):
        return bool(self._result) and bool(self._result[0].sentence)

    def result(self):
        return self._result


def DecodeNumpy(decoder: BrainSpeechDecoder, logits, log_priors, blank_penalty: float):
    """logp = log_softmax(logits) - log_priors; logp[:,0] -= blank_penalty; decoder.Decode(logp)  (lm_decoder.cc:14-37)."""
    lg = torch.as_tensor(np.ascontiguousarray(logits), dtype=torch.float32).to(decoder.device)
    pr = torch.as_tensor(np.ascontiguousarray(log_priors), dtype=torch.float32).to(decoder.device)
    if lg.dim() != 2 or pr.shape != lg.shape:
        raise ValueError("logits and log_priors must be [T, C] arrays of the same shape")
    out = torch.empty_like(lg)
    with torch.cuda.device(decoder.device):
        N.check(N.load().b2t_lm_prologue_f32(ops._p(lg), ops._p(pr), float(blank_penalty), ops._p(out), lg.shape[0],
                                             lg.shape[1], ops._stream()), "b2t_lm_prologue_f32")
    decoder.Decode(out)


def DecodeNumpyLogProbs(decoder: BrainSpeechDecoder, log_probs):
    decoder.Decode(log_probs)
