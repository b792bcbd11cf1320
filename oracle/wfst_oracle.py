"""CPU restatement of the reference's WFST decoder — TEST INFRASTRUCTURE (nothing under nejm-brain-to-text_amd/ imports it).

Parity status: UNPINNED by the reference.  The C++ decoder cannot be built in this image (OpenFST, glog, gflags and
libtorch-1.13.1 are fetched by its CMake from the network, language_model/runtime/server/x86/CMakeLists.txt:33-94) and the
checkout holds no TLG graph, logits fixture or expected n-best for this path; the only reference-authored known answer of
the decoder directory is the prefix-beam table (pinned in oracle/b2t_oracle.py).  What follows restates, function by
function, with the line ranges it follows:

  LatticeFasterDecoder      language_model/runtime/core/kaldi/decoder/lattice-faster-decoder.cc
      InitDecoding :57-75 · FindOrAddToken :250-295 · PruneForwardLinks :297-374 · PruneForwardLinksFinal :380-470 ·
      PruneTokensForFrame :489-514 · PruneActiveTokens :516-545 · ComputeFinalCosts :547-590 · AdvanceDecoding :592-630 ·
      FinalizeDecoding :632-647 · GetCutoff :650-720 · ProcessEmitting :722-824 · ProcessNonemitting :839-909 ·
      GetRawLattice :106-186 · best-path traceback lattice-faster-online-decoder.cc:58-150
  lattice -> n-best         GetLattice :193-213 (DeterminizeLatticePruned: one path per word sequence, the best one) +
                            fst::ShortestPath(nbest) as called at ctc_wfst_beam_search.cc:138-143
  CtcWfstBeamSearch         language_model/runtime/core/decoder/ctc_wfst_beam_search.cc  Search :70-121 (blank-frame
                            skipping, partial best path) · FinalizeSearch :123-160 · ConvertToInputs :162-188 ·
                            DecodableTensorScaled::LogLikelihood :27-33
  BrainSpeechDecoder        brain_speech_decoder.cc:113-137 UpdateResult (scores, word strings) · Rescore / LatticeRescore :47-101
                            (rescore_by_definition, grammar_min_cost below: by enumeration, small lattices only)

The graph is data: CSR arrays (row, ilabel, olabel, weight, next, final) as nejm-brain-to-text_amd/wfst.DecodeGraph holds them.
"""
import heapq
import math

import numpy as np

INF = float("inf")
F32 = np.float32


class Config:
    """LatticeFasterDecoderConfig defaults (lattice-faster-decoder.h:62-72) + CtcWfstBeamSearchOptions
    (ctc_wfst_beam_search.h:55-62); production values: language-model-standalone.py:486-496."""

    def __init__(self, beam=16.0, max_active=2 ** 31 - 1, min_active=200, lattice_beam=10.0, prune_interval=25,
                 beam_delta=0.5, prune_scale=0.1, length_penalty=0.0, acoustic_scale=1.0, nbest=10, blank_skip_thresh=0.98,
                 cutoff_rule="sequential"):
        self.beam, self.max_active, self.min_active, self.lattice_beam = beam, max_active, min_active, lattice_beam
        self.prune_interval, self.beam_delta, self.prune_scale, self.length_penalty = prune_interval, beam_delta, prune_scale, length_penalty
        self.acoustic_scale, self.nbest, self.blank_skip_thresh = acoustic_scale, nbest, blank_skip_thresh
        # "sequential": ProcessEmitting as the reference runs it -- next_cutoff tightens while the hash list is walked, so the
        # tokens created beyond the frame's final cutoff depend on the list's order (HashList below).  "final": the data-parallel
        # rule of csrc/wfst.hip -- every candidate is compared with the frame's FINAL next_cutoff (best candidate +
        # adaptive_beam; the reference reaches the same value at the end of its walk), no token beyond it exists.  The two
        # differ only through GetCutoff's token COUNT / k-th cost in the next frame, i.e. while max_active (or min_active) binds.
        assert cutoff_rule in ("sequential", "final")
        self.cutoff_rule = cutoff_rule


class Token:
    __slots__ = ("tot_cost", "extra_cost", "links", "backpointer", "state")

    def __init__(self, tot_cost, extra_cost, backpointer, state):
        self.tot_cost, self.extra_cost, self.links, self.backpointer, self.state = tot_cost, extra_cost, [], backpointer, state


class Link:
    __slots__ = ("next_tok", "ilabel", "olabel", "graph_cost", "acoustic_cost")

    def __init__(self, next_tok, ilabel, olabel, graph_cost, acoustic_cost):
        self.next_tok, self.ilabel, self.olabel, self.graph_cost, self.acoustic_cost = next_tok, ilabel, olabel, graph_cost, acoustic_cost


class HashList:
    """kaldi/util/hash-list-inl.h restated for what the decoder observes of it: the ORDER in which GetList() / Clear() hand the
    elements out.  Insert (:137-172) puts a key into bucket `key % hash_size_`; an unoccupied bucket is linked at the TAIL of the
    element list (its bucket becomes bucket_list_tail_), an element of an occupied bucket goes behind that bucket's last element:
    the list is the buckets in order of their first occupation, inside a bucket the elements in insertion order.  ProcessEmitting
    walks that list while it tightens next_cutoff (lattice-faster-decoder.cc:786-810), so which over-the-cutoff tokens get
    created depends on it; ProcessNonemitting seeds its LIFO queue from it (:861-865).  SetSize (:37-43) only ever grows the
    table and is called on the empty list (PossiblyResizeHash, lattice-faster-decoder.cc:216-222: num_toks * hash_ratio)."""

    def __init__(self, size):
        self.size, self.buckets = size, {}      # bucket index -> [(key, value)]; dict order = order of first occupation

    def set_size(self, size):
        assert not self.buckets
        self.size = size

    def get(self, key):
        for k, v in self.buckets.get(key % self.size, ()):
            if k == key:
                return v
        return None

    def insert(self, key, val):
        self.buckets.setdefault(key % self.size, []).append((key, val))

    def items(self):
        return [kv for b in self.buckets.values() for kv in b]

    def values(self):
        return [v for b in self.buckets.values() for _, v in b]

    def clear(self):
        out = self.values()
        self.buckets = {}
        return out


class LatticeFasterDecoder:
    HASH_RATIO = 2.0        # LatticeFasterDecoderConfig::hash_ratio (lattice-faster-decoder.h:70)

    def __init__(self, graph, cfg: Config):
        self.g, self.cfg = graph, cfg
        self.row, self.il, self.ol, self.w, self.nx = (np.asarray(getattr(graph, k)) for k in ("row", "ilabel", "olabel", "weight", "next"))
        self.final = np.asarray(graph.final)
        self.has_eps = np.asarray(graph.n_eps) > 0

    # ---- :57-75
    def init_decoding(self):
        # toks_: state -> Token in HashList order; SetSize(1000) in the constructor (:37-38).  (The reference keeps one decoder
        # object, whose table only grows, across utterances; every utterance here starts from the constructor's size.)
        self.toks = HashList(1000)
        self.active = [[]]              # active_toks_[f]: tokens of frame f
        self.must_prune_links, self.must_prune_toks = [True], [True]
        self.cost_offsets = []
        self.finalized = False
        self.final_costs = {}
        start = Token(F32(0.0), F32(0.0), None, int(self.g.start))
        self.active[0].append(start)
        self.toks.insert(int(self.g.start), start)
        self.process_nonemitting(self.cfg.beam)

    def num_frames_decoded(self):
        return len(self.active) - 1

    # ---- :250-295
    def find_or_add_token(self, state, frame_plus_one, tot_cost, backpointer):
        tok = self.toks.get(state)
        if tok is None:
            tok = Token(tot_cost, F32(0.0), backpointer, state)
            self.active[frame_plus_one].append(tok)
            self.toks.insert(state, tok)
            return tok, True
        if tok.tot_cost > tot_cost:
            tok.tot_cost, tok.backpointer = tot_cost, backpointer
            return tok, True
        return tok, False

    # ---- :650-720
    def get_cutoff(self, toks):
        cfg = self.cfg
        costs = np.array([t.tot_cost for t in toks], dtype=np.float32)
        best_i = int(np.argmin(costs)) if len(costs) else -1
        best = costs[best_i] if len(costs) else F32(INF)
        beam_cutoff = best + F32(cfg.beam)
        if cfg.max_active == 2 ** 31 - 1 and cfg.min_active == 0:
            return beam_cutoff, F32(cfg.beam), best_i
        srt = np.sort(costs)            # nth_element leaves the k-th smallest at position k
        max_active_cutoff = srt[cfg.max_active] if len(costs) > cfg.max_active else F32(INF)
        if max_active_cutoff < beam_cutoff:
            return max_active_cutoff, max_active_cutoff - best + F32(cfg.beam_delta), best_i
        min_active_cutoff = F32(INF)
        if len(costs) > cfg.min_active:
            min_active_cutoff = best if cfg.min_active == 0 else srt[cfg.min_active]
        if min_active_cutoff > beam_cutoff:
            return min_active_cutoff, min_active_cutoff - best + F32(cfg.beam_delta), best_i
        return beam_cutoff, F32(cfg.beam), best_i

    # ---- :722-824   loglike[i] = acoustic_scale * logp[i]  (DecodableTensorScaled, ctc_wfst_beam_search.cc:27-33)
    def process_emitting(self, loglike):
        cfg = self.cfg
        frame = len(self.active) - 1
        self.active.append([]); self.must_prune_links.append(True); self.must_prune_toks.append(True)
        final_toks = self.toks.clear()          # Clear(): the element list in HashList order
        cur_cutoff, adaptive_beam, best_i = self.get_cutoff(final_toks)
        new_sz = int(F32(len(final_toks)) * F32(self.HASH_RATIO))     # PossiblyResizeHash :216-222
        if new_sz > self.toks.size:
            self.toks.set_size(new_sz)
        next_cutoff = F32(INF)
        cost_offset = F32(0.0)
        lp = F32(cfg.length_penalty)
        if best_i >= 0:
            tok = final_toks[best_i]
            cost_offset = -tok.tot_cost
            for a in range(self.row[tok.state], self.row[tok.state + 1]):
                if self.il[a] != 0:
                    nw = self.w[a] + cost_offset - loglike[self.il[a] - 1] + tok.tot_cost
                    if tok.state != self.nx[a]:
                        nw = nw + lp
                    if nw + adaptive_beam < next_cutoff:
                        next_cutoff = nw + adaptive_beam
        self.cost_offsets.append(cost_offset)
        if cfg.cutoff_rule == "final":      # the frame's final cutoff first (csrc/wfst.hip pass A), then every candidate against it
            for tok in final_toks:
                if tok.tot_cost <= cur_cutoff:
                    for a in range(self.row[tok.state], self.row[tok.state + 1]):
                        if self.il[a] != 0:
                            graph_cost = self.w[a] + lp if tok.state != self.nx[a] else self.w[a]
                            tot = tok.tot_cost + (cost_offset - loglike[self.il[a] - 1]) + graph_cost
                            if tot + adaptive_beam < next_cutoff:
                                next_cutoff = tot + adaptive_beam
        for tok in final_toks:
            if tok.tot_cost <= cur_cutoff:
                for a in range(self.row[tok.state], self.row[tok.state + 1]):
                    if self.il[a] != 0:
                        ac_cost = cost_offset - loglike[self.il[a] - 1]
                        graph_cost = self.w[a]
                        if tok.state != self.nx[a]:
                            graph_cost = graph_cost + lp
                        tot = tok.tot_cost + ac_cost + graph_cost
                        if tot >= next_cutoff:
                            continue
                        elif tot + adaptive_beam < next_cutoff:
                            next_cutoff = tot + adaptive_beam
                        nt, _ = self.find_or_add_token(int(self.nx[a]), frame + 1, tot, tok)
                        tok.links.insert(0, Link(nt, int(self.il[a]), int(self.ol[a]), graph_cost, ac_cost))
        return next_cutoff

    # ---- :839-909
    def process_nonemitting(self, cutoff):
        frame = len(self.active) - 2
        queue = [t for t in self.toks.values() if self.has_eps[t.state]]
        while queue:
            tok = queue.pop()
            cur = tok.tot_cost
            if cur >= cutoff:
                continue
            tok.links = []
            for a in range(self.row[tok.state], self.row[tok.state + 1]):
                if self.il[a] == 0:
                    graph_cost = self.w[a]
                    tot = cur + graph_cost
                    if tot < cutoff:
                        nt, changed = self.find_or_add_token(int(self.nx[a]), frame + 1, tot, tok)
                        tok.links.insert(0, Link(nt, 0, int(self.ol[a]), graph_cost, F32(0.0)))
                        if changed and self.has_eps[nt.state]:
                            queue.append(nt)

    # ---- :297-374
    def prune_forward_links(self, f, delta):
        extra_changed = links_pruned = False
        changed = True
        while changed:
            changed = False
            for tok in self.active[f]:
                tok_extra = F32(INF)
                kept = []
                for l in tok.links:
                    nt = l.next_tok
                    lec = nt.extra_cost + ((tok.tot_cost + l.acoustic_cost + l.graph_cost) - nt.tot_cost)
                    if lec > self.cfg.lattice_beam:
                        links_pruned = True
                    else:
                        if lec < 0.0:
                            lec = F32(0.0)
                        if lec < tok_extra:
                            tok_extra = lec
                        kept.append(l)
                tok.links = kept
                with np.errstate(invalid="ignore"):
                    differs = abs(tok_extra - tok.extra_cost) > delta     # inf - inf = nan compares false, as in C++
                if differs:
                    changed = True
                tok.extra_cost = tok_extra
            if changed:
                extra_changed = True
        return extra_changed, links_pruned

    # ---- :547-590
    def compute_final_costs(self):
        final_costs, best, best_wf = {}, INF, INF
        for state, tok in self.toks.items():
            fc = float(self.final[state])
            best = min(best, float(tok.tot_cost)); best_wf = min(best_wf, float(tok.tot_cost) + fc)
            if fc != INF:
                final_costs[id(tok)] = F32(fc)
        final_best = best_wf if best_wf != INF else best
        return final_costs, final_best

    # ---- :380-470
    def prune_forward_links_final(self):
        f = len(self.active) - 1
        self.final_costs, self.final_best_cost = self.compute_final_costs()
        self.finalized = True
        self.toks_final = self.toks
        self.toks = HashList(self.toks_final.size)
        changed, delta = True, 1.0e-05
        while changed:
            changed = False
            for tok in self.active[f]:
                fc = F32(0.0) if not self.final_costs else self.final_costs.get(id(tok), F32(INF))
                tok_extra = tok.tot_cost + fc - F32(self.final_best_cost)
                kept = []
                for l in tok.links:
                    nt = l.next_tok
                    lec = nt.extra_cost + ((tok.tot_cost + l.acoustic_cost + l.graph_cost) - nt.tot_cost)
                    if lec > self.cfg.lattice_beam:
                        continue
                    if lec < 0.0:
                        lec = F32(0.0)
                    if lec < tok_extra:
                        tok_extra = lec
                    kept.append(l)
                tok.links = kept
                if tok_extra > self.cfg.lattice_beam:
                    tok_extra = F32(INF)
                a, b = float(tok.extra_cost), float(tok_extra)
                # ApproxEqual (kaldi/base/kaldi-math.h:265-273): equal values (incl. both infinite) are equal; an infinite or
                # NaN difference is NOT (without this rule a token going 0 -> inf passed as unchanged, inf <= delta * inf, and an
                # epsilon link into it survived its destination: tests/test_gpu_wfst.py::test_deep_epsilon_fans_equal_the_oracle)
                diff = abs(a - b) if a != b else 0.0
                if not (a == b or (diff != INF and diff == diff and diff <= delta * (abs(a) + abs(b)))):
                    changed = True
                tok.extra_cost = tok_extra

    # ---- :489-514
    def prune_tokens_for_frame(self, f):
        self.active[f] = [t for t in self.active[f] if t.extra_cost != INF]

    # ---- :516-545
    def prune_active_tokens(self, delta):
        cur = self.num_frames_decoded()
        for f in range(cur - 1, -1, -1):
            if self.must_prune_links[f]:
                ec, lp = self.prune_forward_links(f, delta)
                if ec and f > 0:
                    self.must_prune_links[f - 1] = True
                if lp:
                    self.must_prune_toks[f] = True
                self.must_prune_links[f] = False
            if f + 1 < cur and self.must_prune_toks[f + 1]:
                self.prune_tokens_for_frame(f + 1)
                self.must_prune_toks[f + 1] = False

    # ---- :592-630 (one frame)
    def advance(self, loglike):
        if self.num_frames_decoded() % self.cfg.prune_interval == 0:
            self.prune_active_tokens(self.cfg.lattice_beam * self.cfg.prune_scale)
        cutoff = self.process_emitting(loglike)
        self.process_nonemitting(cutoff)

    # ---- :632-647
    def finalize_decoding(self):
        last = self.num_frames_decoded()
        self.prune_forward_links_final()
        for f in range(last - 1, -1, -1):
            self.prune_forward_links(f, 0.0)
            self.prune_tokens_for_frame(f + 1)
        self.prune_tokens_for_frame(0)

    # ---- lattice-faster-online-decoder.cc:58-150: best token of the last frame, traced back through backpointers
    def best_path(self, use_final_probs):
        final_costs = self.final_costs if self.finalized else (self.compute_final_costs()[0] if use_final_probs else {})
        best, best_tok, best_fc = INF, None, 0.0
        for tok in self.active[-1]:
            cost, fc = float(tok.tot_cost), 0.0
            if use_final_probs and final_costs:
                if id(tok) in final_costs:
                    fc = float(final_costs[id(tok)]); cost += fc
                else:
                    cost = INF
            if cost < best:
                best, best_tok, best_fc = cost, tok, fc
        if best_tok is None:
            return None
        arcs = []          # (ilabel, olabel, graph, acoustic) from the end backwards
        tok, t = best_tok, self.num_frames_decoded() - 1
        while tok.backpointer is not None:
            bl, bc = None, INF
            for l in tok.backpointer.links:
                if l.next_tok is tok:
                    c = float(l.graph_cost) + float(l.acoustic_cost)
                    if c < bc:
                        bl, bc = l, c
            ac = float(bl.acoustic_cost)
            if bl.ilabel != 0:
                ac -= float(self.cost_offsets[t]); t -= 1
            arcs.append((bl.ilabel, bl.olabel, float(bl.graph_cost), ac))
            tok = tok.backpointer
        arcs.reverse()
        alignment = [a[0] for a in arcs if a[0] != 0]
        words = [a[1] for a in arcs if a[1] != 0]
        return alignment, words, sum(a[2] for a in arcs) + best_fc, sum(a[3] for a in arcs)

    # ---- :106-186
    def raw_lattice(self):
        """states = surviving tokens; arcs (src, ilabel, olabel, graph, acoustic - cost_offset, dst); finals with cost."""
        ids, n = {}, 0
        for f, toks in enumerate(self.active):
            for t in toks:
                ids[id(t)] = n; n += 1
        arcs = [[] for _ in range(n)]
        finals = {}
        last = len(self.active) - 1
        for f, toks in enumerate(self.active):
            for t in toks:
                s = ids[id(t)]
                for l in t.links:
                    off = float(self.cost_offsets[f]) if l.ilabel != 0 else 0.0
                    arcs[s].append((l.ilabel, l.olabel, float(l.graph_cost), float(l.acoustic_cost) - off, ids[id(l.next_tok)]))
                if f == last:
                    if self.final_costs:
                        if id(t) in self.final_costs:
                            finals[s] = float(self.final_costs[id(t)])
                    else:
                        finals[s] = 0.0
        return arcs, finals, ids[id(self.active[0][0])]


def nbest_word_sequences(arcs, finals, start, nbest, beam):
    """DeterminizeLatticePruned + ShortestPath(nbest) (lattice-faster-decoder.cc:193-213, ctc_wfst_beam_search.cc:138-143)
    by their definition: the `nbest` cheapest DISTINCT word sequences of the (acyclic) raw lattice within `beam` of the
    best, each with the (graph, acoustic) cost and the input-label alignment of its best path.  Subset construction over
    the word labels -- a determinised state is {lattice state: best (total, graph, acoustic, alignment)} -- explored
    best-first with the exact backward cost as the bound, so that results come out in order of cost."""
    n = len(arcs)
    # beta[s]: cheapest cost from s to a final state incl. the final cost (all arc costs are >= 0: Dijkstra backwards)
    rev = [[] for _ in range(n)]
    for s in range(n):
        for il, ol, g, a, d in arcs[s]:
            rev[d].append((s, g + a))
    beta = [INF] * n
    heap = [(c, s) for s, c in finals.items()]
    for c, s in heap:
        beta[s] = c
    heapq.heapify(heap)
    while heap:
        c, s = heapq.heappop(heap)
        if c > beta[s]:
            continue
        for p, w in rev[s]:
            if c + w < beta[p]:
                beta[p] = c + w
                heapq.heappush(heap, (c + w, p))
    if beta[start] == INF:
        return []
    limit = beta[start] + beam + 1e-4

    def closure(sub):
        # extend over arcs without a word (olabel 0), keeping the best entry per lattice state
        heap = [(v[0], s) for s, v in sub.items()]
        heapq.heapify(heap)
        while heap:
            c, s = heapq.heappop(heap)
            if c > sub[s][0]:
                continue
            tot, gr, ac, ali = sub[s]
            for il, ol, g, a, d in arcs[s]:
                if ol == 0 and tot + g + a + beta[d] <= limit:
                    nt = tot + g + a
                    if d not in sub or nt < sub[d][0]:
                        sub[d] = (nt, gr + g, ac + a, ali + ((il,) if il else ()))
                        heapq.heappush(heap, (nt, d))
        return sub

    results, counter = [], 0
    start_sub = closure({start: (0.0, 0.0, 0.0, ())})
    pq = [(beta[start], 0, 0, (), start_sub)]       # (bound, kind 0 = subset / 1 = finished sequence, tie, words, payload)
    while pq and len(results) < int(nbest):
        bound, kind, _, words, payload = heapq.heappop(pq)
        if bound > limit:
            break
        if kind == 1:
            results.append(payload)
            continue
        sub = payload
        fin = None
        for s, (tot, gr, ac, ali) in sub.items():
            if s in finals:
                c = (tot + finals[s], gr + finals[s], ac, words, ali)
                if fin is None or c[0] < fin[0]:
                    fin = c
        if fin is not None and fin[0] <= limit:
            counter += 1
            heapq.heappush(pq, (fin[0], 1, counter, words, fin))
        by_word = {}
        for s, (tot, gr, ac, ali) in sub.items():
            for il, ol, g, a, d in arcs[s]:
                if ol != 0 and tot + g + a + beta[d] <= limit:
                    nt = tot + g + a
                    tgt = by_word.setdefault(ol, {})
                    if d not in tgt or nt < tgt[d][0]:
                        tgt[d] = (nt, gr + g, ac + a, ali + ((il,) if il else ()))
        for ol, tgt in by_word.items():
            sub2 = closure(tgt)
            b = min(v[0] + beta[s] for s, v in sub2.items())
            counter += 1
            heapq.heappush(pq, (b, 0, counter, words + (ol,), sub2))
    return results


def grammar_min_cost(arcs, finals, start, words, backoff):
    """What composing ONE word sequence with a grammar and determinising costs (brain_speech_decoder.cc:47-58: fst::Compose with the
    grammar read by ReadAndPrepareLmFst -- back-off arcs are epsilons, so a route may take them anywhere --, then
    DeterminizeLattice keeps the cheapest route): min over routes of arc weights + the final cost.  arcs[s] = [(ilabel, weight,
    next)], finals {state: cost}; arcs labelled `backoff` are the free ones.  inf = not accepted."""
    def close(d):
        stack = list(d)
        while stack:
            s = stack.pop()
            for il, w, n in arcs[s]:
                if il == backoff and d[s] + w < d.get(n, INF):
                    d[n] = d[s] + w
                    stack.append(n)
        return d
    cur = close({start: 0.0})
    for word in words:
        nxt = {}
        for s, c in cur.items():
            for il, w, n in arcs[s]:
                if il == word and c + w < nxt.get(n, INF):
                    nxt[n] = c + w
        if not nxt:
            return INF
        cur = close(nxt)
    return min([c + finals[s] for s, c in cur.items() if s in finals] or [INF])


def rescore_by_definition(arcs, finals, start, g_old, g_new, backoff, nbest, beam):
    """BrainSpeechDecoder::Rescore (brain_speech_decoder.cc:61-101) by enumeration of a SMALL acyclic lattice.
    lat_ (ctc_wfst_beam_search.cc:138-141: GetLattice = determinised and pruned with lattice_beam) holds ONE path per word sequence,
    the cheapest, for the sequences within the beam.  LatticeRescore(-1) (:47-58): graph := -graph; compose with the old grammar;
    DeterminizeLattice keeps the cheapest (path, route) per word sequence = -graph + min-route G_old; graph := -graph again, i.e.
    graph - G_old(W).  LatticeRescore(+1) adds min-route G_new(W).  ShortestPath(n) on graph + acoustic.
    arcs[s] = [(ilabel, olabel, graph, acoustic, next)] as for nbest_word_sequences; g_old / g_new = (arcs, finals, start) for
    grammar_min_cost.  Returns [(words, graph', acoustic, alignment)] best first."""
    best = {}
    stack = [(start, (), (), 0.0, 0.0)]
    while stack:
        s, words, ali, g, a = stack.pop()
        if s in finals:
            t = (g + finals[s] + a, g + finals[s], a, ali)
            if words not in best or t[0] < best[words][0]:
                best[words] = t
        for il, ol, gr, ac, n in arcs[s]:
            stack.append((n, words + ((ol,) if ol else ()), ali + ((il,) if il else ()), g + gr, a + ac))
    if not best:
        return []
    cut = min(t[0] for t in best.values()) + beam
    res = []
    for words, (tot, g, a, ali) in best.items():
        if tot > cut:
            continue
        go, gn = grammar_min_cost(*g_old, list(words), backoff), grammar_min_cost(*g_new, list(words), backoff)
        if go == INF or gn == INF:
            continue
        res.append((words, g - go + gn, a, ali))
    res.sort(key=lambda e: e[1] + e[2])
    return res[:nbest]


class CtcWfstBeamSearch:
    def __init__(self, graph, cfg: Config):
        self.cfg = cfg
        self.dec = LatticeFasterDecoder(graph, cfg)
        self.reset()

    def reset(self):
        self.num_frames, self.mapping = 0, []
        self.is_last_frame_blank, self.last_best, self.last_frame_prob = False, 0, None
        self.inputs, self.outputs, self.likelihood, self.times = [], [], [], []
        self.dec.init_decoding()

    # ---- ctc_wfst_beam_search.cc:70-121
    def search(self, logp):
        logp = np.asarray(logp, dtype=np.float32)
        sc = F32(self.cfg.acoustic_scale)
        for i in range(logp.shape[0]):
            blank_score = math.exp(float(logp[i, 0]))
            if blank_score > self.cfg.blank_skip_thresh:
                self.is_last_frame_blank, self.last_frame_prob = True, logp[i]
            else:
                cur_best = int(np.argmax(logp[i]))
                if cur_best != 0 and self.is_last_frame_blank and cur_best == self.last_best:
                    self.dec.advance(sc * self.last_frame_prob)
                    self.mapping.append(self.num_frames - 1)
                self.last_best = cur_best
                self.dec.advance(sc * logp[i])
                self.mapping.append(self.num_frames)
                self.is_last_frame_blank = False
            self.num_frames += 1
        self.inputs, self.outputs, self.likelihood = [], [], []
        if self.mapping:
            ali, words, gc, ac = self.dec.best_path(False)
            self.inputs, self.outputs, self.likelihood = [self.convert_to_inputs(ali)[0]], [words], [(-gc, -ac)]

    # ---- :123-160
    def finalize_search(self):
        self.dec.finalize_decoding()
        self.inputs, self.outputs, self.likelihood, self.times = [], [], [], []
        if not self.mapping:
            return
        if self.cfg.nbest == 1:
            ali, words, gc, ac = self.dec.best_path(True)
            entries = [(gc + ac, gc, ac, tuple(words), tuple(ali))]
        else:
            arcs, finals, start = self.dec.raw_lattice()
            entries = nbest_word_sequences(arcs, finals, start, self.cfg.nbest, self.cfg.lattice_beam)
        for tot, gc, ac, words, ali in entries:
            inp, tm = self.convert_to_inputs(list(ali))
            self.inputs.append(inp); self.outputs.append(list(words)); self.likelihood.append((-gc, -ac)); self.times.append(tm)

    # ---- :162-188 (reads past the end of the alignment are treated as "different label")
    def convert_to_inputs(self, alignment):
        inp, tm, cur, n = [], [], 0, len(alignment)
        while cur < n:
            while cur < n and alignment[cur] - 1 == 0:
                cur += 1
            while cur + 1 < n and alignment[cur + 1] == alignment[cur]:
                cur += 1
            if cur < n:
                inp.append(alignment[cur] - 1); tm.append(self.mapping[cur]); cur += 1
        return inp, tm


def process_blank(s):
    """language_model/runtime/core/utils/string.cc:121-146."""
    out = []
    for ch in s:
        if ch not in ("▁", " "):
            out.append(ch)
        elif out and out[-1] != " ":
            out.append(" ")
    return "".join(out).rstrip(" ").lower()


def decode_results(search: CtcWfstBeamSearch, words):
    """brain_speech_decoder.cc:113-137 UpdateResult: [(sentence, ac_score, lm_score)]."""
    res = []
    for hyp, (lm, ac) in zip(search.outputs, search.likelihood):
        sent = process_blank("".join(" " + words[w] for w in hyp))
        res.append((sent, ac / search.cfg.acoustic_scale, lm))
    return res
