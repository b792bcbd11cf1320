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
from typing import Dict, List, Sequence, Tuple

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

    def to_device(self, device):
        import torch
        if self._dev is None or self._dev["row"].device != torch.device(device):
            self._dev = {k: torch.from_numpy(getattr(self, k)).to(device) for k in ("row", "ilabel", "olabel", "weight", "next", "n_eps", "final")}
        return self._dev

    def nbytes(self):
        return sum(getattr(self, k).nbytes for k in ("row", "ilabel", "olabel", "weight", "next", "n_eps", "final"))


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
    return DecodeGraph(read_openfst_vector(fst_path), words)


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
