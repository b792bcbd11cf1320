"""Token-level back-off n-gram language model for the fused CTC prefix beam search (b2t_prefix_beam_search_lm_f32).

The reference scores hypotheses with SRILM/ARPA n-gram models compiled into a word-level WFST
(language_model/runtime/core/kaldi/lm/arpa-lm-compiler.cc:162-285 reads the ARPA file; the graph is then searched by
language_model/runtime/core/decoder/ctc_wfst_beam_search.cc).  Neither the LM files nor the graph tooling are part of
the checkout, so this module implements the part that is well defined without them: an ARPA model over the decoder's
own output tokens (the 41 phoneme classes of evaluate_model_helpers.LOGIT_TO_PHONEME), laid out for the GPU as a
back-off automaton resident in HBM:

  node      one per n-gram (w1..wk) of the ARPA file, node 0 = empty context
  child     [n_nodes][V] int32   child[n][w] = node of (n-gram of n) + w, or -1            (dense: V = C + 3 <= 67)
  logp      [n_nodes]   float32  ln p(wk | w1..wk-1)
  bow       [n_nodes]   float32  ln back-off weight of the n-gram used as a context
  suffix    [n_nodes]   int32    node of the longest proper suffix that exists (back-off target), root for unigrams
  nstate    [n_nodes]   int32    LM state after emitting the n-gram: the longest suffix of it of length <= order-1 that exists

so that p(w | state s) is at most `order` dependent steps:  c = child[s][w];  found -> logp[c], next state nstate[c];
else add bow[s] and continue from suffix[s]; at the root an unseen word scores <unk> (or `unk_logp`).
Vocabulary: class ids 0..C-1 (the blank's entry is never queried), then <s> = C, </s> = C+1, <unk> = C+2.
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence, Tuple

import numpy as np

LN10 = math.log(10.0)


def parse_arpa(text: str) -> Tuple[int, Dict[Tuple[str, ...], Tuple[float, float]]]:
    """ARPA text -> (order, {n-gram: (log10 p, log10 back-off)})."""
    table: Dict[Tuple[str, ...], Tuple[float, float]] = {}
    order = cur = 0
    for raw in text.splitlines():
        line = raw.strip()
        if not line or line == "\\data\\" or line.startswith("ngram "):
            continue
        if line == "\\end\\":
            break
        if line.startswith("\\") and line.endswith("-grams:"):
            cur = int(line[1:line.index("-")])
            order = max(order, cur)
            continue
        parts = line.split()
        if cur == 0 or len(parts) < cur + 1:
            continue
        words = tuple(parts[1:1 + cur])
        table[words] = (float(parts[0]), float(parts[1 + cur]) if len(parts) > 1 + cur else 0.0)
    if order == 0:
        raise ValueError("not an ARPA language model (no \\N-grams: sections)")
    return order, table


class NGramLM:
    """Back-off automaton of one ARPA model over the decoder's token set (see module docstring)."""

    def __init__(self, order: int, table: Dict[Tuple[str, ...], Tuple[float, float]], id_to_word: Sequence[str],
                 unk_logp: float = -99.0 * LN10):
        C = len(id_to_word)
        self.C, self.V, self.order = C, C + 3, order
        self.bos, self.eos, self.unk = C, C + 1, C + 2
        wid = {w: i for i, w in enumerate(id_to_word) if w is not None}
        wid.update({"<s>": self.bos, "</s>": self.eos, "<unk>": self.unk})
        # n-grams over known words only (an ARPA trained on another token set contributes nothing for foreign words)
        grams = {g: v for g, v in table.items() if all(w in wid for w in g)}
        ids: Dict[Tuple[str, ...], int] = {(): 0}
        for g in sorted(grams, key=len):
            for k in range(1, len(g) + 1):   # every prefix of an n-gram is a node (ARPA guarantees them; be tolerant)
                if g[:k] not in ids:
                    ids[g[:k]] = len(ids)
        n = len(ids)
        child = np.full((n, self.V), -1, dtype=np.int32)
        logp = np.full((n,), unk_logp, dtype=np.float32)
        bow = np.zeros((n,), dtype=np.float32)
        suffix = np.zeros((n,), dtype=np.int32)
        nstate = np.zeros((n,), dtype=np.int32)
        depth = np.zeros((n,), dtype=np.int32)
        for g, i in ids.items():
            if not g:
                continue
            child[ids[g[:-1]], wid[g[-1]]] = i
            depth[i] = len(g)
            if g in grams:
                logp[i] = grams[g][0] * LN10
                bow[i] = grams[g][1] * LN10
            s = g[1:]
            while s not in ids:
                s = s[1:]
            suffix[i] = ids[s]
            st = g if len(g) <= order - 1 else g[1:]
            while st not in ids:
                st = st[1:]
            nstate[i] = ids[st]
        # an n-gram that is only a prefix of longer ones (no probability of its own) must not be "found": unlink it
        for g, i in ids.items():
            if g and g not in grams:
                logp[i] = np.float32(np.nan)
        self.child, self.logp, self.bow, self.suffix, self.nstate, self.depth = child, logp, bow, suffix, nstate, depth
        self.has_entry = np.array([bool(g in grams) or not g for g in ids], dtype=bool)
        for g, i in ids.items():   # missing own entry -> treat as absent child (back off through it)
            if g and g not in grams:
                child[ids[g[:-1]], wid[g[-1]]] = -1
        self.start_state = int(child[0, self.bos]) if child[0, self.bos] >= 0 and order > 1 else 0
        self.unk_logp = float(logp[child[0, self.unk]]) if child[0, self.unk] >= 0 else float(unk_logp)
        self.n_nodes = n
        self._dev = None

    @classmethod
    def from_arpa(cls, text: str, id_to_word: Sequence[str], **kw) -> "NGramLM":
        order, table = parse_arpa(text)
        return cls(order, table, id_to_word, **kw)

    # host-side scoring with the same automaton (used by the wrapper for argument checks and by tests)
    def step(self, state: int, w: int) -> Tuple[float, int]:
        acc = 0.0
        s = state
        while True:
            c = int(self.child[s, w])
            if c >= 0:
                return acc + float(self.logp[c]), int(self.nstate[c])
            if s == 0:
                return acc + self.unk_logp, 0
            acc += float(self.bow[s])
            s = int(self.suffix[s])

    def sentence_logp(self, ids: Sequence[int], bos: bool = True, eos: bool = False) -> float:
        s = self.start_state if bos else 0
        tot = 0.0
        for w in list(ids) + ([self.eos] if eos else []):
            lp, s = self.step(s, int(w))
            tot += lp
        return tot

    def to_device(self, device):
        import torch
        if self._dev is None or self._dev["device"] != str(device):
            t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
            self._dev = dict(device=str(device), child=t(self.child.reshape(-1)), logp=t(self.logp), bow=t(self.bow),
                             suffix=t(self.suffix), nstate=t(self.nstate))
        return self._dev


def synthetic_arpa(id_to_word: Sequence[str], order: int, n_per_order: int, seed: int = 0, skip: Sequence[int] = (0,)) -> str:
    """A random but well-formed ARPA model over the given tokens (bench / tests: the real LMs are not in the checkout).
    Every unigram is present; higher orders hold `n_per_order` random n-grams whose prefixes all exist."""
    rng = np.random.RandomState(seed)
    words = [w for i, w in enumerate(id_to_word) if i not in skip and w is not None]
    levels: List[Dict[Tuple[str, ...], Tuple[float, float]]] = []
    uni = {(w,): (float(np.log10(p)), float(-rng.uniform(0.1, 0.8)))
           for w, p in zip(words + ["</s>", "<unk>"], rng.dirichlet(np.ones(len(words) + 2) * 2.0))}
    uni[("<s>",)] = (-99.0, float(-rng.uniform(0.1, 0.8)))
    levels.append(uni)
    for k in range(2, order + 1):
        prev = [g for g in levels[-1] if g[-1] != "</s>"]
        cur: Dict[Tuple[str, ...], Tuple[float, float]] = {}
        tries = 0
        while len(cur) < n_per_order and tries < 20 * n_per_order:
            tries += 1
            g = prev[rng.randint(len(prev))] + (words[rng.randint(len(words))] if rng.rand() > 0.05 else "</s>",)
            if g[0] == "</s>" or g in cur:
                continue
            cur[g] = (float(-rng.uniform(0.05, 2.5)), float(-rng.uniform(0.05, 0.9)) if k < order else 0.0)
        levels.append(cur)
    out = ["\\data\\"] + [f"ngram {k + 1}={len(l)}" for k, l in enumerate(levels)] + [""]
    for k, l in enumerate(levels):
        out.append(f"\\{k + 1}-grams:")
        for g, (lp, bw) in l.items():
            out.append(f"{lp:.6f}\t{' '.join(g)}" + (f"\t{bw:.6f}" if k + 1 < order else ""))
        out.append("")
    out.append("\\end\\")
    return "\n".join(out)


# ----------------------------------------------------------------------------------------------------------------
# Word-level decoding: pronunciation lexicon trie + sparse word n-gram automaton (b2t_prefix_beam_search_lex_f32)
# ----------------------------------------------------------------------------------------------------------------
class Lexicon:
    """Pronunciation trie over the decoder's classes.  Words are delimited by the SIL class in the token stream (the
    reference's phoneme transcriptions put SIL between words; its lexicon FST is built with "SIL" as the optional
    silence, tools/fst/ctc_compile_dict_token.sh:94-98).  Arrays for the GPU:
      child [n_nodes][C] int32 (-1 = no edge), wbeg/wend [n_nodes] -> ranges of `wlist` (word ids ending at the node).
    Word ids are the ranks of the words in sorted order (homophone ties are broken by that order)."""

    def __init__(self, prons: Dict[str, Sequence[Sequence[int]]], n_classes: int):
        self.C = n_classes
        self.words = sorted(prons)
        wid = {w: i for i, w in enumerate(self.words)}
        ids: Dict[Tuple[int, ...], int] = {(): 0}
        ends: Dict[int, List[int]] = {}
        for w in self.words:
            for pron in prons[w]:
                pron = tuple(int(x) for x in pron)
                if not pron or any(c <= 0 or c >= n_classes for c in pron):
                    raise ValueError(f"bad pronunciation for {w!r}: {pron}")
                for k in range(1, len(pron) + 1):
                    if pron[:k] not in ids:
                        ids[pron[:k]] = len(ids)
                ends.setdefault(ids[pron], []).append(wid[w])
        n = len(ids)
        self.child = np.full((n, n_classes), -1, dtype=np.int32)
        for p, i in ids.items():
            if p:
                self.child[ids[p[:-1]], p[-1]] = i
        self.wbeg = np.zeros(n, dtype=np.int32); self.wend = np.zeros(n, dtype=np.int32)
        wl: List[int] = []
        for i in range(n):
            self.wbeg[i] = len(wl)
            wl.extend(sorted(set(ends.get(i, []))))
            self.wend[i] = len(wl)
        self.wlist = np.array(wl if wl else [0], dtype=np.int32)
        self.n_nodes = n
        self._dev = None

    def to_device(self, device):
        import torch
        if self._dev is None or self._dev["device"] != str(device):
            t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
            self._dev = dict(device=str(device), child=t(self.child.reshape(-1)), wbeg=t(self.wbeg), wend=t(self.wend),
                             wlist=t(self.wlist))
        return self._dev


class SparseNGramLM:
    """Back-off automaton over a large (word) vocabulary: like NGramLM, but the children of a node are a sorted array
    (ctok / cnode in [cb[n], ce[n])) searched by bisection instead of a dense table.  Vocabulary: the lexicon's word
    ids 0..W-1, then <s> = W, </s> = W+1, <unk> = W+2."""

    def __init__(self, order: int, table: Dict[Tuple[str, ...], Tuple[float, float]], words: Sequence[str],
                 unk_logp: float = -99.0 * LN10):
        W = len(words)
        self.W, self.order = W, order
        self.bos, self.eos, self.unk = W, W + 1, W + 2
        wid = {w: i for i, w in enumerate(words)}
        wid.update({"<s>": self.bos, "</s>": self.eos, "<unk>": self.unk})
        grams = {tuple(wid[w] for w in g): v for g, v in table.items() if all(w in wid for w in g)}
        ids: Dict[Tuple[int, ...], int] = {(): 0}
        for g in sorted(grams, key=lambda g: (len(g), g)):
            for k in range(1, len(g) + 1):
                if g[:k] not in ids:
                    ids[g[:k]] = len(ids)
        n = len(ids)
        logp = np.full(n, unk_logp, dtype=np.float32); bow = np.zeros(n, dtype=np.float32)
        suffix = np.zeros(n, dtype=np.int32); nstate = np.zeros(n, dtype=np.int32)
        kids: List[List[Tuple[int, int]]] = [[] for _ in range(n)]
        for g, i in ids.items():
            if not g:
                continue
            if g in grams:
                logp[i] = grams[g][0] * LN10; bow[i] = grams[g][1] * LN10
                kids[ids[g[:-1]]].append((g[-1], i))      # only n-grams with an entry of their own can be "found"
            s = g[1:]
            while s not in ids:
                s = s[1:]
            suffix[i] = ids[s]
            st = g if len(g) <= order - 1 else g[1:]
            while st not in ids:
                st = st[1:]
            nstate[i] = ids[st]
        cb = np.zeros(n, dtype=np.int32); ce = np.zeros(n, dtype=np.int32)
        ctok: List[int] = []; cnode: List[int] = []
        for i in range(n):
            cb[i] = len(ctok)
            for tkn, node in sorted(kids[i]):
                ctok.append(tkn); cnode.append(node)
            ce[i] = len(ctok)
        self.cb, self.ce = cb, ce
        self.ctok = np.array(ctok if ctok else [0], dtype=np.int32); self.cnode = np.array(cnode if cnode else [0], dtype=np.int32)
        self.logp, self.bow, self.suffix, self.nstate = logp, bow, suffix, nstate
        self.n_nodes = n
        c = self._find(0, self.bos)
        self.start_state = int(self.nstate[c]) if c >= 0 and order > 1 else 0
        u = self._find(0, self.unk)
        self.unk_logp = float(logp[u]) if u >= 0 else float(unk_logp)
        self._dev = None

    @classmethod
    def from_arpa(cls, text: str, words: Sequence[str], **kw) -> "SparseNGramLM":
        order, table = parse_arpa(text)
        return cls(order, table, words, **kw)

    def _find(self, s: int, w: int) -> int:
        lo, hi = int(self.cb[s]), int(self.ce[s])
        while lo < hi:
            mid = (lo + hi) // 2
            if self.ctok[mid] < w:
                lo = mid + 1
            else:
                hi = mid
        return int(self.cnode[lo]) if lo < int(self.ce[s]) and self.ctok[lo] == w else -1

    def step(self, state: int, w: int) -> Tuple[float, int]:
        acc, s = 0.0, state
        while True:
            c = self._find(s, w)
            if c >= 0:
                return acc + float(self.logp[c]), int(self.nstate[c])
            if s == 0:
                return acc + self.unk_logp, 0
            acc += float(self.bow[s]); s = int(self.suffix[s])

    def to_device(self, device):
        import torch
        if self._dev is None or self._dev["device"] != str(device):
            t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
            self._dev = dict(device=str(device), cb=t(self.cb), ce=t(self.ce), ctok=t(self.ctok), cnode=t(self.cnode),
                             logp=t(self.logp), bow=t(self.bow), suffix=t(self.suffix), nstate=t(self.nstate))
        return self._dev


def replay_words(lex: Lexicon, lm: SparseNGramLM, tokens: Sequence[int], alpha: float, beta: float, sil: int = 1,
                 add_eos: bool = True):
    """Host replay of the kernel's word emissions for ONE phoneme hypothesis: -> (words, lm score) or (None, -inf) if
    the hypothesis ends inside a word.  Same rules, same tie-breaks (best LM probability, then lowest word id)."""
    state, node, words, score = lm.start_state, 0, [], 0.0

    def emit():
        nonlocal state, node, score
        best = None
        for k in range(int(lex.wbeg[node]), int(lex.wend[node])):
            w = int(lex.wlist[k])
            lp, ns = lm.step(state, w)
            if best is None or lp > best[0]:
                best = (lp, ns, w)
        score += alpha * best[0] + beta
        state = best[1]; words.append(lex.words[best[2]]); node = 0
    for c in tokens:
        c = int(c)
        if c == sil:
            if node != 0:
                if lex.wend[node] == lex.wbeg[node]:
                    return None, -math.inf
                emit()
        else:
            node = int(lex.child[node, c])
            if node < 0:
                return None, -math.inf
    if node != 0:
        if lex.wend[node] == lex.wbeg[node]:
            return None, -math.inf
        emit()
    if add_eos:
        score += alpha * lm.step(state, lm.eos)[0]
    return words, score


def synthetic_lexicon(n_words: int, n_classes: int, seed: int = 0, sil: int = 1, blank: int = 0):
    """Random pronunciation dictionary (bench / tests): words w00000.. with 2-8 phonemes, a few homophones."""
    rng = np.random.RandomState(seed)
    phones = [c for c in range(n_classes) if c not in (sil, blank)]
    prons: Dict[str, List[Tuple[int, ...]]] = {}
    seen: List[Tuple[int, ...]] = []
    for i in range(n_words):
        if seen and rng.rand() < 0.03:
            p = seen[rng.randint(len(seen))]          # homophone
        else:
            p = tuple(int(phones[j]) for j in rng.randint(len(phones), size=rng.randint(2, 9)))
            seen.append(p)
        prons[f"w{i:05d}"] = [p]
    return prons


def synthetic_word_arpa(words: Sequence[str], order: int, n_per_order: int, seed: int = 0) -> str:
    return synthetic_arpa([None] + list(words), order, n_per_order, seed=seed, skip=(0,))
