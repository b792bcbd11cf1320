"""Decode-graph construction: pronunciation lexicon + ARPA n-gram -> T o L o G held as CSR arc arrays in HBM.

The reference builds TLG.fst offline with OpenFST / Kaldi binaries that are not in this image
(language_model/tools/fst/make_tlg.sh:29-46); this module restates the recipe's semantics over a small in-memory FST type:

  T  token FST, CTC topology     tools/fst/ctc_token_fst_corrected.py:42-57 ("decode" mode), tokens.txt order
                                 <eps> <blk> SIL units.. #0.. (tools/fst/ctc_compile_dict_token.sh:65)
  L  lexicon FST                 tools/fst/make_lexicon_fst.pl (optional silence with probability sil_prob, pronunciation
                                 probabilities 1.0) + the #0 self-loops of fstaddselfloops (kaldi/fstext/pre-determinize-inl.h:657)
  G  grammar FST                 kaldi/lm/arpa-lm-compiler.cc:162-285 (one state per history, back-off arcs, highest-order
                                 n-grams go straight to the back-off history), then eps2disambig.pl (back-off ilabel -> #0),
                                 s2eps.pl (<s>, </s> -> <eps>) and fstrmepsilon, as make_tlg.sh:29-40 pipes them
  TLG = T o (L o G)              make_tlg.sh:43-46; epsilon-filtered composition

Not reproduced: fstdeterminizestar / fstminimizeencoded of L o G (and therefore the lexicon disambiguation symbols
#1.. that only exist to make that determinisation possible).  They change the SIZE of the graph, not the cost of any
(input sequence, word sequence) pair, so best paths and n-best lists over the graph are the same.  Costs are natural-log
(-ln p), as arpa2fst produces them.
"""
from __future__ import annotations

import math
from collections import deque
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

LN10 = math.log(10.0)
EPS = 0


class Fst:
    """Arcs (src, ilabel, olabel, weight, dst) in the tropical semiring; `final` maps state -> final cost."""

    def __init__(self):
        self.n = 0
        self.start = -1
        self.arcs: List[Tuple[int, int, int, float, int]] = []
        self.final: Dict[int, float] = {}

        self._csr = None

    def add_state(self) -> int:
        self.n += 1
        return self.n - 1

    def add_arc(self, s, il, ol, w, d):
        self.arcs.append((s, il, ol, float(w), d))
        self._csr = None

    def out(self):
        o: List[List[Tuple[int, int, float, int]]] = [[] for _ in range(self.n)]
        for s, il, ol, w, d in self.arcs:
            o[s].append((il, ol, w, d))
        return o

    def csr(self):
        """(row[n+1], ilabel, olabel, weight, next) with the arcs of a state sorted by ilabel; built once (numpy) and
        cached until the next add_arc: what grammar_score walks (one bisection per (state, label))."""
        if self._csr is None or self._csr[5] != len(self.arcs):
            a = np.array(self.arcs, dtype=np.float64).reshape(-1, 5)
            src, il = a[:, 0].astype(np.int64), a[:, 1].astype(np.int64)
            order = np.lexsort((il, src))
            row = np.zeros(self.n + 1, dtype=np.int64)
            np.add.at(row, src + 1, 1)
            self._csr = (np.cumsum(row), il[order], a[order, 2].astype(np.int64), a[order, 3], a[order, 4].astype(np.int64),
                         len(self.arcs))
        return self._csr[:5]


# ------------------------------------------------------------------------------------------------
def token_fst(n_units: int, disambig_tokens: Sequence[int]) -> Fst:
    """T in "decode" mode.  n_units = SIL + phonemes (tokens 2 .. n_units+1); token 1 = <blk>.  State 0 = blank/start,
    state i = inside unit i.  ilabel = token id (the decodable reads logp[ilabel - 1], ctc_wfst_beam_search.cc:27-33)."""
    f = Fst()
    for _ in range(n_units + 1):
        f.add_state()
    f.start = 0
    il = lambda n: n + 1
    f.add_arc(0, il(0), EPS, 0.0, 0)
    for i in range(1, n_units + 1):
        f.add_arc(0, il(i), il(i), 0.0, i)       # enter unit i, emit it
        f.add_arc(i, il(i), EPS, 0.0, i)         # repeat
        f.add_arc(i, il(0), EPS, 0.0, 0)         # blank
    for i in range(1, n_units + 1):
        for j in range(1, n_units + 1):
            if i != j:
                f.add_arc(i, il(j), il(j), 0.0, j)
    for i in range(n_units + 1):
        f.final[i] = 0.0
        for t in disambig_tokens:
            f.add_arc(i, EPS, t, 0.0, i)
    return f


def lexicon_fst(prons: Dict[str, Sequence[Sequence[int]]], word_id: Dict[str, int], sil_prob: float, sil_token: int,
                tok_disambig0: int, word_disambig0: int) -> Fst:
    """L per make_lexicon_fst.pl with optional silence (the variant without a silence disambiguation symbol).
    prons: word -> pronunciations as TOKEN ids (SIL = 2, phonemes 3..)."""
    f = Fst()
    if not (0.0 < sil_prob < 1.0):
        raise ValueError("sil_prob must be in (0, 1)")
    silcost, nosilcost = -math.log(sil_prob), -math.log(1.0 - sil_prob)
    start, loop, sil = f.add_state(), f.add_state(), f.add_state()
    f.start = start
    f.add_arc(start, EPS, EPS, nosilcost, loop)
    f.add_arc(start, sil_token, EPS, silcost, loop)
    f.add_arc(sil, sil_token, EPS, 0.0, loop)
    for w in sorted(prons):
        for pron in prons[w]:
            pron = list(pron)
            if not pron:
                raise ValueError(f"empty pronunciation for {w}")
            s, wo = loop, word_id[w]
            for k, p in enumerate(pron):
                if k + 1 < len(pron):
                    ns = f.add_state()
                    f.add_arc(s, p, wo, 0.0, ns)
                    wo, s = EPS, ns
                elif p != sil_token:
                    f.add_arc(s, p, wo, nosilcost, loop)
                    f.add_arc(s, p, wo, silcost, sil)
                else:                                  # no point putting optional silence after a silence word
                    f.add_arc(s, p, wo, 0.0, loop)
    f.final[loop] = 0.0
    # fstaddselfloops: #0:#0 on every state that is final or has an arc with a word on it
    needs = set(f.final)
    for s, il, ol, w, d in f.arcs:
        if ol != EPS:
            needs.add(s)
    for s in sorted(needs):
        f.add_arc(s, tok_disambig0, word_disambig0, 0.0, s)
    return f


def parse_arpa(text: str):
    """ARPA text -> (order, [(words tuple, log10 p, log10 back-off)] in file order)."""
    grams, order, cur = [], 0, 0
    for raw in text.splitlines():
        line = raw.strip()
        if not line or line == "\\data\\" or line.startswith("ngram "):
            continue
        if line == "\\end\\":
            break
        if line.startswith("\\") and line.endswith("-grams:"):
            cur = int(line[1:line.index("-")]); order = max(order, cur)
            continue
        parts = line.split()
        if cur == 0 or len(parts) < cur + 1:
            continue
        # make_tlg.sh:29-34 greps these lines away before arpa2fst sees them
        joined = " ".join(parts[1:1 + cur])
        if "<s> <s>" in joined or "</s> <s>" in joined or "</s> </s>" in joined or "<unk>" in joined.lower():
            continue
        grams.append((tuple(parts[1:1 + cur]), float(parts[0]), float(parts[1 + cur]) if len(parts) > 1 + cur else 0.0))
    if order == 0:
        raise ValueError("not an ARPA language model")
    return order, grams


def grammar_fst(arpa_text: str, word_id: Dict[str, int], word_disambig0: int) -> Fst:
    """G: arpa2fst --keep-symbols (ArpaLmCompilerImpl::ConsumeNGram, sub_eps = 0, <s> / </s> kept) | eps2disambig.pl |
    s2eps.pl | fstrmepsilon.  Words missing from word_id drop their n-grams (arpa2fst's OOV handling skips them)."""
    order, grams = parse_arpa(arpa_text)
    f = Fst()
    hist: Dict[Tuple[str, ...], int] = {(): f.add_state()}
    eos_state = f.add_state()
    f.final[eos_state] = 0.0
    BOS, EOS = "<s>", "</s>"
    start_arc = None                      # (start, <s>-history state)
    eos_arcs = []                         # (source, cost): the </s> arcs, turned into final costs by s2eps + rmepsilon

    def backoff_target(key):
        while key not in hist:
            key = key[1:]
        return hist[key]

    def state_with_backoff(key, backoff_cost):
        if key in hist:
            return hist[key]
        d = f.add_state()
        hist[key] = d
        f.add_arc(d, word_disambig0, EPS, backoff_cost, backoff_target(key[1:]))   # eps2disambig: ilabel #0, olabel <eps>
        return d

    for words, lp, bo in grams:
        if any(w not in word_id and w not in (BOS, EOS) for w in words):
            continue
        heads = words[:-1]
        if heads not in hist:
            continue                      # "skipped: no parent (n-1)-gram exists"
        src, sym = hist[heads], words[-1]
        cost = -lp * LN10
        is_highest = len(words) == order
        if sym == EOS:
            eos_arcs.append((src, cost))
            continue
        dest = state_with_backoff(words[1:] if is_highest else words, -bo * LN10)
        if sym == BOS:
            start_arc = dest              # accepting <s> is free; s2eps + rmepsilon make its destination the start state
            continue
        f.add_arc(src, word_id[sym], word_id[sym], cost, dest)
    if start_arc is None:
        raise ValueError("the ARPA model has no <s> unigram")
    f.start = start_arc
    for src, cost in eos_arcs:
        f.final[src] = min(f.final.get(src, math.inf), cost)
    del f.final[eos_state]
    return f


# ------------------------------------------------------------------------------------------------
def compose(a: Fst, b: Fst) -> Fst:
    """a o b with the epsilon-matching filter (filter state 0: free, 1: a moved alone on an output epsilon, 2: b moved
    alone on an input epsilon), so that an epsilon path is generated once.  Only pairs reachable from the start exist."""
    ao, bo = a.out(), b.out()
    b_by_il = [dict() for _ in range(b.n)]
    for s in range(b.n):
        for il, ol, w, d in bo[s]:
            b_by_il[s].setdefault(il, []).append((ol, w, d))
    out = Fst()
    ids: Dict[Tuple[int, int, int], int] = {}

    def sid(k):
        if k not in ids:
            ids[k] = out.add_state()
            queue.append(k)
        return ids[k]

    # a's arcs by output label, with their position in the state's arc list: a state of T has an arc per token while a
    # state of L o G accepts one or two, so the labels are matched from the smaller side and the matches are emitted in a's
    # arc order (the order, and with it the state numbering, of walking all of a's arcs and probing b for each -- which was
    # 42 M dictionary probes for a 3000-word T o (L o G))
    a_by_ol = [dict() for _ in range(a.n)]
    a_eps = [[] for _ in range(a.n)]
    for sa_ in range(a.n):
        for idx, (il, ol, w, d) in enumerate(ao[sa_]):
            if ol != EPS:
                a_by_ol[sa_].setdefault(ol, []).append((idx, il, w, d))
            else:
                a_eps[sa_].append((idx, il, w, d))
    queue: deque = deque()
    out.start = sid((a.start, b.start, 0))
    while queue:
        k = queue.popleft()
        sa, sb, fs = k
        s = ids[k]
        if sa in a.final and sb in b.final:
            out.final[s] = a.final[sa] + b.final[sb]
        am, bb = a_by_ol[sa], b_by_il[sb]
        items = []
        if len(bb) <= len(am):
            for lab, lst in bb.items():
                alist = am.get(lab)
                if alist:
                    for t in alist:
                        items.append((t, lst))
        else:
            for lab, alist in am.items():
                lst = bb.get(lab)
                if lst:
                    for t in alist:
                        items.append((t, lst))
        for t in a_eps[sa]:
            items.append((t, None))
        if len(items) > 1:
            items.sort(key=lambda it: it[0][0])
        beps = bb.get(EPS, ())
        for (idx, il, w, d), lst in items:
            if lst is not None:
                for ol2, w2, d2 in lst:
                    out.add_arc(s, il, ol2, w + w2, sid((d, d2, 0)))
            else:
                if fs != 2:                                   # a alone
                    out.add_arc(s, il, EPS, w, sid((d, sb, 1)))
                if fs == 0:                                   # both on epsilon
                    for ol2, w2, d2 in beps:
                        out.add_arc(s, il, ol2, w + w2, sid((d, d2, 0)))
        if fs != 1:                                           # b alone
            for ol2, w2, d2 in b_by_il[sb].get(EPS, ()):
                out.add_arc(s, EPS, ol2, w2, sid((sa, d2, 2)))
    return out


def trim(f: Fst) -> Fst:
    """Keep states that are reachable from the start AND can reach a final state; renumber (start = 0)."""
    fwd: List[List[int]] = [[] for _ in range(f.n)]
    bwd: List[List[int]] = [[] for _ in range(f.n)]
    for s, il, ol, w, d in f.arcs:
        fwd[s].append(d); bwd[d].append(s)

    def reach(seeds, adj):
        seen = set(seeds); st = list(seeds)
        while st:
            u = st.pop()
            for v in adj[u]:
                if v not in seen:
                    seen.add(v); st.append(v)
        return seen

    keep = reach([f.start], fwd) & reach(list(f.final), bwd)
    if f.start not in keep:
        raise ValueError("the graph accepts nothing")
    order = [f.start] + sorted(keep - {f.start})
    new = {s: i for i, s in enumerate(order)}
    g = Fst()
    g.n, g.start = len(order), 0
    g.arcs = [(new[s], il, ol, w, new[d]) for s, il, ol, w, d in f.arcs if s in keep and d in keep]
    g.final = {new[s]: c for s, c in f.final.items() if s in keep}
    return g


class DecodeGraph:
    """TLG as CSR arrays, arcs of a state sorted by ilabel (input-epsilon arcs first):
    row[S+1], ilabel / olabel / next int32, weight f32, n_eps[S] (number of input-epsilon arcs of the state), final[S] f32
    (inf = not final).  `words` maps olabel -> word."""

    def __init__(self, f: Fst, words: Sequence[str]):
        arcs = sorted(f.arcs, key=lambda a: (a[0], a[1], a[2], a[4], a[3]))
        self.n_states, self.n_arcs = f.n, len(arcs)
        src = np.array([a[0] for a in arcs], dtype=np.int64)
        self.row = np.zeros(f.n + 1, dtype=np.int32)
        np.add.at(self.row, src + 1, 1)
        self.row = np.cumsum(self.row).astype(np.int32)
        self.ilabel = np.array([a[1] for a in arcs], dtype=np.int32)
        self.olabel = np.array([a[2] for a in arcs], dtype=np.int32)
        self.weight = np.array([a[3] for a in arcs], dtype=np.float32)
        self.next = np.array([a[4] for a in arcs], dtype=np.int32)
        self.n_eps = np.zeros(f.n, dtype=np.int32)
        np.add.at(self.n_eps, src[self.ilabel == 0], 1)
        self.final = np.full(f.n, np.inf, dtype=np.float32)
        for s, c in f.final.items():
            self.final[s] = c
        self.start = f.start
        self.words = list(words)
        self._dev = None

    compact = False      # set_compact(True): 10-byte arcs on the device (b2t_wfst_graph_t.compact)

    def set_compact(self, on: bool = True):
        """Arcs as {labels = ilabel | olabel << 7, weight as IEEE half, next}: 10 bytes instead of 16 (the graph of the reference's
        vocabulary: 1.77 -> 1.23 GB).  Weights lose precision (relative error <= 2^-11 = 4.9e-4: ~0.01 on an LM cost of 20)."""
        on = bool(on)
        if on:
            if self.ilabel.size and (int(self.ilabel.max()) > 127 or int(self.olabel.max()) >= 1 << 25):
                raise ValueError("compact arcs need ilabel <= 127 and olabel < 2^25")
            if self.weight.size and float(np.abs(self.weight[np.isfinite(self.weight)]).max(initial=0.0)) > 65000.0:
                raise ValueError("compact arcs: a weight does not fit IEEE half")
        if on != self.compact:
            self.compact, self._dev = on, None
        return self

    def half_rounded(self) -> "DecodeGraph":
        """A full-width copy whose weights are what the compact form stores (for the exact parity test)."""
        import copy
        g = copy.copy(self)
        g.weight = self.weight.astype(np.float16).astype(np.float32)
        g.compact, g._dev = False, None
        return g

    def to_device(self, device):
        import torch
        if self._dev is None or self._dev["row"].device != torch.device(device):
            if self.compact:
                labels = (self.ilabel.astype(np.uint32) | (self.olabel.astype(np.uint32) << np.uint32(7))).view(np.int32)
                arrays = dict(row=self.row, labels=labels, weight_f16=self.weight.astype(np.float16).view(np.int16), next=self.next,
                              n_eps=self.n_eps, final=self.final)
            else:
                arrays = {k: getattr(self, k) for k in ("row", "ilabel", "olabel", "weight", "next", "n_eps", "final")}
            self._dev = {k: torch.from_numpy(np.ascontiguousarray(v)).to(device) for k, v in arrays.items()}
        return self._dev

    def nbytes(self):
        per_arc = 10 if self.compact else 16
        return self.n_arcs * per_arc + sum(getattr(self, k).nbytes for k in ("row", "n_eps", "final"))


def build_tlg(prons: Dict[str, Sequence[Sequence[int]]], arpa_text: str, n_classes: int = 41, sil_prob: float = 0.5,
              sil_class: int = 1) -> DecodeGraph:
    """prons: word -> pronunciations as decoder CLASS ids (the LM decoder's order [BLANK, SIL, AA .. ZH],
    evaluate_model_helpers.py:79-83; class c is token c + 1).  Returns T o L o G trimmed, as a DecodeGraph."""
    words = sorted(prons)
    table = ["<eps>"] + words + ["#0", "<s>", "</s>"]            # words.txt (ctc_compile_dict_token.sh:72-86)
    word_id = {w: i for i, w in enumerate(table) if 0 < i <= len(words)}
    wd0 = len(words) + 1
    n_units = n_classes - 1                                       # SIL + phonemes
    td0 = n_units + 2                                             # tokens.txt: <eps> <blk> SIL units.. #0
    tok_prons = {w: [[int(c) + 1 for c in p] for p in ps] for w, ps in prons.items()}
    T = token_fst(n_units, [td0])
    L = lexicon_fst(tok_prons, word_id, sil_prob, sil_class + 1, td0, wd0)
    G = grammar_fst(arpa_text, word_id, wd0)
    LG = compose(L, G)
    TLG = trim(compose(T, LG))
    if any(il > n_classes for _, il, _, _, _ in TLG.arcs):
        raise AssertionError("a disambiguation token survived as an input label")
    return DecodeGraph(TLG, table)


# ------------------------------------------------------------------------------------------------
# OpenFST "vector" container (what fstcompile / fsttablecompose write for TLG.fst): header, then per state the final
# weight and its arcs.  Layout per OpenFST 1.6 (fst/fst.h FstHeader::Write, fst/vector-fst.h VectorFstImpl::Read/Write):
#   int32 magic 2125659606 | string fst_type "vector" | string arc_type "standard" | int32 version (2) | int32 flags |
#   uint64 properties | int64 start | int64 num_states | int64 num_arcs | [symbol tables if flags & 3] |
#   per state: float final_weight, int64 n_arcs, n_arcs x (int32 ilabel, int32 olabel, float weight, int32 nextstate)
# (strings are int32 length + bytes).  No TLG.fst ships with the reference checkout, so this reader is exercised by the
# round trip with write_openfst_vector only.
# ------------------------------------------------------------------------------------------------
_FST_MAGIC = 2125659606


def write_openfst_vector(f: Fst, path: str):
    import struct
    with open(path, "wb") as fh:
        def wstr(s):
            b = s.encode(); fh.write(struct.pack("<i", len(b))); fh.write(b)
        fh.write(struct.pack("<i", _FST_MAGIC)); wstr("vector"); wstr("standard")
        fh.write(struct.pack("<iiQqqq", 2, 0, 0, f.start, f.n, len(f.arcs)))
        out = f.out()
        for s in range(f.n):
            fh.write(struct.pack("<fq", f.final.get(s, math.inf), len(out[s])))
            for il, ol, w, d in out[s]:
                fh.write(struct.pack("<iifi", il, ol, w, d))


def read_openfst_vector(path: str) -> Fst:
    import struct
    with open(path, "rb") as fh:
        data = fh.read()
    pos = 0

    def rd(fmt):
        nonlocal pos
        v = struct.unpack_from("<" + fmt, data, pos); pos += struct.calcsize("<" + fmt)
        return v

    def rstr():
        nonlocal pos
        (n,) = rd("i"); s = data[pos:pos + n].decode(); pos += n
        return s

    (magic,) = rd("i")
    if magic != _FST_MAGIC:
        raise ValueError(f"{path}: not an OpenFST binary (magic {magic})")
    ftype, atype = rstr(), rstr()
    if ftype != "vector" or atype != "standard":
        raise ValueError(f"{path}: fst type '{ftype}' / arc type '{atype}' not supported (need vector / standard: fstconvert --fst_type=vector)")
    version, flags, props, start, ns, na = rd("iiQqqq")
    if flags & 3:
        raise ValueError(f"{path}: embedded symbol tables are not supported (the recipe compiles with --keep_isymbols=false)")
    f = Fst()
    f.n, f.start = int(ns), int(start)
    for s in range(f.n):
        fw, n = rd("fq")
        if fw != math.inf:
            f.final[s] = fw
        for _ in range(n):
            il, ol, w, d = rd("iifi")
            f.arcs.append((s, il, ol, w, d))
    return f


def graph_from_files(fst_path: str, dict_path: str) -> DecodeGraph:
    """TLG.fst (OpenFST vector/standard binary, or an .npz written by DecodeGraph.save) + words.txt -> DecodeGraph."""
    words: List[str] = []
    if dict_path:
        table = {}
        with open(dict_path) as fh:
            for line in fh:
                p = line.split()
                if len(p) >= 2:
                    table[int(p[-1])] = p[0]
        words = [table.get(i, str(i)) for i in range(max(table) + 1)] if table else []
    if fst_path.endswith(".npz"):
        z = np.load(fst_path, allow_pickle=False)
        g = DecodeGraph.__new__(DecodeGraph)
        for k in ("row", "ilabel", "olabel", "weight", "next", "n_eps", "final"):
            setattr(g, k, z[k])
        g.n_states, g.n_arcs, g.start = int(g.row.shape[0] - 1), int(g.ilabel.shape[0]), int(z["start"])
        g.words = words or [str(w) for w in z["words"]]
        g._dev = None
        return g
    # the OpenFST container is parsed and arc-sorted in C++ (b2t_fst_read_openfst): no Python object per arc
    return HostFst.read_openfst(fst_path).arcsort().to_graph(words)


def save_graph(g: DecodeGraph, path: str):
    np.savez(path, row=g.row, ilabel=g.ilabel, olabel=g.olabel, weight=g.weight, next=g.next, n_eps=g.n_eps, final=g.final,
             start=np.int64(g.start), words=np.array(g.words))


def grammar_score(G: Fst, word_ids: Sequence[int], backoff_label: int):
    """Cheapest path through G accepting the word sequence (back-off arcs carry `backoff_label` on the input side and
    may be taken freely) plus the final cost: the LM cost a lattice path picks up when composed with G
    (BrainSpeechDecoder::LatticeRescore, brain_speech_decoder.cc:44-58).  Walks G's cached CSR form: the arcs of a state
    are sorted by ilabel, so a (state, label) lookup is one bisection, whatever the size of the grammar."""
    row, il, ol, wt, nx = G.csr()

    def arcs(s, label):
        a, e = int(row[s]), int(row[s + 1])
        lo = a + int(np.searchsorted(il[a:e], label, "left"))
        hi = a + int(np.searchsorted(il[a:e], label, "right"))
        return range(lo, hi)

    def close(d):
        st = list(d)
        while st:
            s = st.pop()
            for k in arcs(s, backoff_label):
                n_, c = int(nx[k]), d[s] + float(wt[k])
                if n_ not in d or c < d[n_]:
                    d[n_] = c; st.append(n_)
        return d

    cur = close({G.start: 0.0})
    for wid in word_ids:
        nxt = {}
        for s, c in cur.items():
            for k in arcs(s, int(wid)):
                n_, c2 = int(nx[k]), c + float(wt[k])
                if n_ not in nxt or c2 < nxt[n_]:
                    nxt[n_] = c2
        if not nxt:
            return math.inf
        cur = close(nxt)
    return min((c + G.final[s] for s, c in cur.items() if s in G.final), default=math.inf)


# ------------------------------------------------------------------------------------------------
# Native graph compiler (csrc/graphc.cpp through the C ABI): the same algebra on flat arrays, for graphs of the reference's
# size class, plus the two optimisation passes of make_tlg.sh:43-44 that the Python path above leaves out.
# ------------------------------------------------------------------------------------------------
class HostFst:
    """Handle to a host FST of libb2t_hip.so (b2t_fst_*: CSR arrays in C++).  Operations return new HostFst objects."""

    def __init__(self, handle):
        if not handle:
            import b2t_native as N
            raise RuntimeError("graph compiler: " + N.last_error())
        self._h = handle

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                import b2t_native as N
                N.load().b2t_fst_free(self._h)
                self._h = None
        except Exception:
            pass

    @staticmethod
    def _lib():
        import b2t_native as N
        return N.load()

    @classmethod
    def from_arrays(cls, n_states, start, src, il, ol, w, dst, final_cost):
        import ctypes as C
        A = [np.ascontiguousarray(x, dtype=t) for x, t in ((src, np.int32), (il, np.int32), (ol, np.int32), (w, np.float32), (dst, np.int32))]
        fc = np.ascontiguousarray(final_cost, dtype=np.float32)
        assert fc.shape[0] == n_states
        P = lambda a: a.ctypes.data_as(C.c_void_p)
        return cls(cls._lib().b2t_fst_from_arrays(int(n_states), int(start), len(A[0]), P(A[0]), P(A[1]), P(A[2]), P(A[3]), P(A[4]), P(fc)))

    @classmethod
    def from_fst(cls, f: Fst):
        a = np.array(f.arcs, dtype=np.float64).reshape(-1, 5)
        fc = np.full(f.n, np.inf, np.float32)
        for s, c in f.final.items():
            fc[s] = c
        return cls.from_arrays(f.n, f.start, a[:, 0], a[:, 1], a[:, 2], a[:, 3], a[:, 4], fc)

    @classmethod
    def read_openfst(cls, path: str):
        import os
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        return cls(cls._lib().b2t_fst_read_openfst(path.encode()))

    def write_openfst(self, path: str):
        import b2t_native as N
        N.check(self._lib().b2t_fst_write_openfst(self._h, path.encode()), "b2t_fst_write_openfst")

    def info(self):
        import b2t_native as N
        out = (N.LL * 4)()
        N.check(self._lib().b2t_fst_info(self._h, out), "b2t_fst_info")
        return dict(n_states=int(out[0]), n_arcs=int(out[1]), start=int(out[2]), n_final=int(out[3]))

    def arrays(self):
        """(row int64 [n+1], ilabel, olabel, weight, next, final_cost)"""
        import ctypes as C
        import b2t_native as N
        i = self.info()
        row = np.zeros(i["n_states"] + 1, np.int64)
        il, ol, nx = (np.zeros(max(1, i["n_arcs"]), np.int32) for _ in range(3))
        w = np.zeros(max(1, i["n_arcs"]), np.float32); fc = np.zeros(i["n_states"], np.float32)
        P = lambda a: a.ctypes.data_as(C.c_void_p)
        N.check(self._lib().b2t_fst_to_arrays(self._h, P(row), P(il), P(ol), P(w), P(nx), P(fc)), "b2t_fst_to_arrays")
        n = i["n_arcs"]
        return row, il[:n], ol[:n], w[:n], nx[:n], fc

    def compose(self, other: "HostFst"):
        return HostFst(self._lib().b2t_fst_compose(self._h, other._h))

    def trim(self):
        return HostFst(self._lib().b2t_fst_trim(self._h))

    def determinize_star(self, use_log: bool = True, delta: float = 1.0 / 1024, max_states: int = 0):
        import ctypes as C
        return HostFst(self._lib().b2t_fst_determinize_star(self._h, int(use_log), C.c_float(delta), int(max_states)))

    def minimize_encoded(self, delta: float = 1.0 / 1024):
        import ctypes as C
        return HostFst(self._lib().b2t_fst_minimize_encoded(self._h, C.c_float(delta)))

    def arcsort(self, by_olabel: bool = False):
        return HostFst(self._lib().b2t_fst_arcsort(self._h, int(by_olabel)))

    def prepare_lm(self, disambig_id: Optional[int] = None):
        """fst::ReadAndPrepareLmFst (kaldi/fstext/kaldi-fst-io.cc:129-147): a grammar that is not an acceptor is projected on
        its OUTPUT labels (the #0 that eps2disambig.pl put on the back-off arcs' input side becomes the epsilon of their output
        side), an acceptor is left as it is; then arcs are sorted by input label.  Returns (prepared HostFst, back-off label):
        0 after a projection, and for an acceptor 0 too -- unless no arc carries label 0 while arcs carry `disambig_id`
        (words.txt's id of "#0"): a grammar compiled with #0 on both sides, which fst::Compose could not back off through at
        all; following those arcs is the friendlier reading."""
        import ctypes as C
        backoff = C.c_int(0)      # (the acceptor test, the projection and the sort stay in C++: the grammar is never copied into numpy)
        h = self._lib().b2t_fst_prepare_lm(self._h, -1 if disambig_id is None else int(disambig_id), C.byref(backoff))
        return HostFst(h), int(backoff.value)

    def grammar_score(self, word_ids: Sequence[int], backoff_label: int) -> float:
        """Needs an ilabel-sorted grammar (arcsort())."""
        import ctypes as C
        w = np.ascontiguousarray(word_ids, dtype=np.int32)
        return float(self._lib().b2t_fst_grammar_score(self._h, w.ctypes.data_as(C.c_void_p), len(w), int(backoff_label)))

    def to_fst(self) -> Fst:
        row, il, ol, w, nx, fc = self.arrays()
        f = Fst()
        f.n, f.start = len(fc), self.info()["start"]
        src = np.repeat(np.arange(f.n), np.diff(row))
        f.arcs = [(int(s), int(a), int(b), float(c), int(d)) for s, a, b, c, d in zip(src, il, ol, w, nx)]
        f.final = {int(s): float(c) for s, c in enumerate(fc) if np.isfinite(c)}
        return f

    def to_graph(self, words: Sequence[str]) -> "DecodeGraph":
        """DecodeGraph (CSR for the device) of an ilabel-sorted FST, without going through Python tuples."""
        row, il, ol, w, nx, fc = self.arrays()
        g = DecodeGraph.__new__(DecodeGraph)
        g.n_states, g.n_arcs, g.start = len(fc), len(il), self.info()["start"]
        if g.n_arcs >= 2 ** 31:
            raise ValueError("graph too large for 32-bit arc offsets")
        g.row = row.astype(np.int32)
        g.ilabel, g.olabel, g.weight, g.next = il, ol, w, nx
        src = np.repeat(np.arange(g.n_states), np.diff(row))
        if np.any(np.diff(il.astype(np.int64) + src.astype(np.int64) * (1 << 32)) < 0):
            raise ValueError("to_graph needs arcs sorted by ilabel within each state (arcsort())")
        g.n_eps = np.bincount(src[il == 0], minlength=g.n_states).astype(np.int32)
        g.final = fc
        g.words = list(words)
        g._dev = None
        return g


def lex_disambig(prons: Dict[str, Sequence[Sequence[int]]]):
    """tools/fst/add_lex_disambig.pl: a disambiguation symbol #k (k >= 1) after every pronunciation that is shared by several
    entries or is a proper prefix of another, numbered per phone sequence.  Returns ({(word, i): k or 0}, max k).  Entries are
    visited in the order of the sorted word list (the recipe's lexicon is `sort | uniq`-ed)."""
    count: Dict[Tuple[int, ...], int] = {}
    issub = set()
    order = [(w, i, tuple(p)) for w in sorted(prons) for i, p in enumerate(prons[w])]
    for _, _, p in order:
        count[p] = count.get(p, 0) + 1
        for k in range(len(p)):
            issub.add(p[:k])
    last: Dict[Tuple[int, ...], int] = {}
    out, mx = {}, 0
    for w, i, p in order:
        if p not in issub and count[p] == 1:
            out[(w, i)] = 0
            continue
        k = last.get(p, 0) + 1
        last[p] = k
        mx = max(mx, k)
        out[(w, i)] = k
    return out, mx


def lexicon_fst_disambig(prons, word_id, sil_prob, sil_token, tok_disambig0, word_disambig0):
    """L as ctc_compile_dict_token.sh:57-98 builds it: add_lex_disambig.pl, then make_lexicon_fst.pl --pron-probs with optional
    silence AND the silence disambiguation symbol '#'$ndisambig (make_lexicon_fst.pl:101-152), then fstaddselfloops for #0.
    Token of #k = tok_disambig0 + k.  Returns (Fst, number of disambiguation tokens incl. #0)."""
    dis, mx = lex_disambig(prons)
    ndis = mx + 1                              # '#'$ndisambig with ndisambig = max + 1 is the silence disambiguation symbol
    f = Fst()
    if not (0.0 <= sil_prob < 1.0):
        raise ValueError("sil_prob must be in [0, 1)")
    if sil_prob == 0.0:
        # make_lexicon_fst.pl:62-98 (the script's default in ctc_compile_dict_token.sh:22): no optional silence, ONE state
        # that is start, loop and final; the silence disambiguation symbol stays in tokens.txt but is on no arc
        loop = f.add_state()
        f.start = loop
        for w in sorted(prons):
            for i, pron in enumerate(prons[w]):
                seq = list(pron) + ([tok_disambig0 + dis[(w, i)]] if dis[(w, i)] else [])
                s, wo = loop, word_id[w]
                for k, p in enumerate(seq):
                    ns = f.add_state() if k + 1 < len(seq) else loop
                    f.add_arc(s, p, wo, 0.0, ns)
                    wo, s = EPS, ns
    else:
        silcost, nosilcost = -math.log(sil_prob), -math.log(1.0 - sil_prob)
        start, loop, sil, disst = f.add_state(), f.add_state(), f.add_state(), f.add_state()
        f.start = start
        f.add_arc(start, EPS, EPS, nosilcost, loop)
        f.add_arc(start, sil_token, EPS, silcost, disst)
        f.add_arc(sil, sil_token, EPS, 0.0, disst)
        f.add_arc(disst, tok_disambig0 + ndis, EPS, 0.0, loop)
        for w in sorted(prons):
            for i, pron in enumerate(prons[w]):
                seq = list(pron) + ([tok_disambig0 + dis[(w, i)]] if dis[(w, i)] else [])
                s, wo = loop, word_id[w]
                for k, p in enumerate(seq):
                    if k + 1 < len(seq):
                        ns = f.add_state()
                        f.add_arc(s, p, wo, 0.0, ns)
                        wo, s = EPS, ns
                    elif p != sil_token:
                        f.add_arc(s, p, wo, nosilcost, loop)
                        f.add_arc(s, p, wo, silcost, sil)
                    else:
                        f.add_arc(s, p, wo, 0.0, loop)
    f.final[loop] = 0.0
    needs = set(f.final)
    for s, il, ol, w_, d in f.arcs:
        if ol != EPS:
            needs.add(s)
    for s in sorted(needs):
        f.add_arc(s, tok_disambig0, word_disambig0, 0.0, s)
    return f, ndis + 1


def build_tlg_native(prons: Dict[str, Sequence[Sequence[int]]], arpa_text: str, n_classes: int = 41, sil_prob: float = 0.5,
                     sil_class: int = 1, optimize: bool = True, stats: dict = None) -> DecodeGraph:
    """make_tlg.sh:29-46 end to end with the native compiler: L (with lexicon disambiguation symbols) o G, then -- optimize --
    fstdeterminizestar --use-log=true | fstminimizeencoded, fstarcsort, and T o LG.  `stats` (a dict) receives sizes and times."""
    import time
    t0 = time.time()
    words = sorted(prons)
    table = ["<eps>"] + words + ["#0", "<s>", "</s>"]
    word_id = {w: i for i, w in enumerate(table) if 0 < i <= len(words)}
    wd0 = len(words) + 1
    n_units = n_classes - 1
    td0 = n_units + 2
    tok_prons = {w: [[int(c) + 1 for c in p] for p in ps] for w, ps in prons.items()}
    Lf, n_dis = lexicon_fst_disambig(tok_prons, word_id, sil_prob, sil_class + 1, td0, wd0)
    Tf = token_fst(n_units, [td0 + k for k in range(n_dis)])
    Gf = grammar_fst(arpa_text, word_id, wd0)
    t1 = time.time()
    L, T, G = HostFst.from_fst(Lf), HostFst.from_fst(Tf), HostFst.from_fst(Gf).arcsort()
    LG = L.compose(G)
    t2 = time.time()
    raw = LG.info()
    if optimize:
        LG = LG.determinize_star(use_log=True)
        det = LG.info()
        LG = LG.minimize_encoded()
    t3 = time.time()
    opt = LG.info()
    TLG = T.arcsort(by_olabel=True).compose(LG.arcsort()).trim().arcsort()
    t4 = time.time()
    g = TLG.to_graph(table)
    if np.any(g.ilabel > n_classes):
        raise AssertionError("a disambiguation token survived as an input label")
    if stats is not None:
        stats.update(words=len(words), disambig_tokens=n_dis, L=L.info(), G=G.info(), LG_raw=raw, LG=opt,
                     LG_determinized=(det if optimize else None), TLG=TLG.info(), tlg_bytes=g.nbytes(),
                     s_build_LTG_python=round(t1 - t0, 2), s_compose_LG=round(t2 - t1, 2), s_determinize_minimize=round(t3 - t2, 2),
                     s_compose_TLG=round(t4 - t3, 2), s_total=round(time.time() - t0, 2))
    return g
