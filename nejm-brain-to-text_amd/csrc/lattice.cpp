// lattice.cpp — HOST side of FinalizeSearch (ctc_wfst_beam_search.cc:123-160): the n-best distinct word sequences of the
// pruned token lattice the GPU search leaves behind.  The reference gets them from DeterminizeLatticePruned (one path per
// word sequence: the best one; lattice-faster-decoder.cc:193-213) followed by fst::ShortestPath(nbest); this is that
// definition computed directly: subset construction over the word labels (a determinised state is a set of lattice states
// with their best cost so far), expanded best-first with the exact backward cost as the bound, so that sequences come
// out in order of total cost and the search stops after `nbest` of them or at `beam` above the best.
// Runs once per utterance (lattices reach 10^5 arcs); the per-frame hot loop is csrc/wfst.hip.
// A determinised state is kept as two parallel vectors in insertion order (deterministic iteration, unlike a hash map's) and
// membership is a scratch array over the lattice states, marked while one subset is being built: no hashing, no node
// allocations, no subset copies (first version with std::unordered_map subsets: 135 ms for a 175 k-arc lattice, now ~1/4).
#include <algorithm>
#include <queue>
#include <vector>
#include <cmath>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include "common.h"

namespace {

struct Arc { int il, ol, dst; float g, a, d; };         // d: the part of g that a rescoring pass exchanged (0 otherwise)
struct Entry { double tot, gr, ac; int ali; double dl; };   // ali: node of the alignment list (-1 = empty); dl: sum of d so far
struct Node { int parent, label; };
struct Subset { std::vector<int> st; std::vector<Entry> en; };   // lattice states and their best entries, insertion order
struct Trans { int ol, dst; double tot, gr, ac; int ali, il; double dl; };   // a word arc out of a subset

// Work arrays of a call, kept per host thread: a 175 k-arc lattice needs ~7 MB of them, and fresh allocations of that size are
// mapped and unmapped by the allocator on every call (~1700 page faults, a fifth of the call).
struct Scratch {
  std::vector<int> out_off, pending, po, order, rank, slot;
  std::vector<Arc> out_arc;
  std::vector<double> fin, beta, fin_o, beta_o;
};

struct Item {
  double bound; int kind; long long tie; int words; int payload;   // kind 0: subset (payload = index), 1: finished (payload = index)
  bool operator<(const Item& o) const {                             // priority_queue is a max-heap: invert
    if (bound != o.bound) return bound > o.bound;
    if (kind != o.kind) return kind < o.kind;                       // finished sequences before subsets of equal bound
    return tie > o.tie;
  }
};

}  // namespace

// `delta` / `final_delta` (both or neither): the lattice has been through a rescoring exchange (b2t_lattice_rescore_nbest_host
// below) that moved graph[i] by delta[i] and a final cost by final_delta[i].  The answers are then ranked by the costs as given
// (the NEW ones), while `beam` keeps its meaning on the OLD ones (graph - delta): a word sequence takes part iff its best OLD
// path is within `beam` of the best OLD path, which is what the reference's GetLattice left in lat_ before Rescore() ran.
// Precondition: the accumulated delta of a path depends on its word sequence only (true for the product lattice built below).
int b2t_lattice_nbest_core(int n_states, int start, int n_arcs, const int32_t* src, const int32_t* dst,
                           const int32_t* ilabel, const int32_t* olabel, const float* graph, const float* acoustic,
                           const float* delta, int n_final, const int32_t* final_state, const float* final_cost,
                           const float* final_delta, int nbest, float beam,
                           int32_t* out_words, int32_t* w_off, int w_cap, int32_t* out_ali, int32_t* a_off, int a_cap,
                           float* costs) {
  if (n_states <= 0 || start < 0 || start >= n_states || n_arcs < 0 || nbest <= 0 || !w_off || !a_off || !costs) {
    b2t::set_error("lattice_nbest: bad arguments");
    return -1;
  }
  const double INF = INFINITY;
  const auto tm_in = std::chrono::steady_clock::now();
  // CSR adjacency (lattices reach 10^5 arcs: no per-state vectors).  Only the forward direction: the backward costs below are
  // relaxed over a state's OUTGOING arcs in reverse topological order, so no transposed copy is built (setup was 40 % of the
  // call for a 175 k-arc lattice with both directions).
  static thread_local Scratch tls;
  std::vector<int>& out_off = tls.out_off; std::vector<int>& pending = tls.pending;
  out_off.assign(n_states + 1, 0); pending.assign(n_states, 0);
  for (int i = 0; i < n_arcs; ++i) {
    if (src[i] < 0 || src[i] >= n_states || dst[i] < 0 || dst[i] >= n_states) { b2t::set_error("lattice_nbest: arc %d out of range", i); return -1; }
    ++out_off[src[i] + 1]; ++pending[dst[i]];
  }
  for (int s = 0; s < n_states; ++s) out_off[s + 1] += out_off[s];
  const auto ts1 = std::chrono::steady_clock::now();
  std::vector<Arc>& out_arc = tls.out_arc;
  if ((int)out_arc.size() < n_arcs) out_arc.resize(n_arcs);
  {
    std::vector<int>& po = tls.po;
    po.assign(out_off.begin(), out_off.end() - 1);
    for (int i = 0; i < n_arcs; ++i)                          // arc order within a state = input order (deterministic ties)
      out_arc[po[src[i]]++] = Arc{ilabel[i], olabel[i], dst[i], graph[i], acoustic[i], delta ? delta[i] : 0.f};
  }
  const auto ts2 = std::chrono::steady_clock::now();
  std::vector<double>& fin = tls.fin; std::vector<double>& beta = tls.beta;
  fin.assign(n_states, INF); beta.assign(n_states, INF);
  for (int i = 0; i < n_final; ++i) fin[final_state[i]] = std::min(fin[final_state[i]], (double)final_cost[i]);
  const bool resc = delta != nullptr;
  if (resc && !final_delta) { b2t::set_error("lattice_nbest: delta without final_delta"); return -1; }
  std::vector<double>& fin_o = tls.fin_o; std::vector<double>& beta_o = tls.beta_o;   // the OLD costs' counterparts (rescoring only)
  if (resc) {
    fin_o.assign(n_states, INF); beta_o.assign(n_states, INF);
    for (int i = 0; i < n_final; ++i) fin_o[final_state[i]] = std::min(fin_o[final_state[i]], (double)final_cost[i] - (double)final_delta[i]);
  }
  // beta: cheapest completion incl. the final cost.  The lattice is acyclic, so one sweep in reverse topological order (Kahn,
  // O(states + arcs)) gives it EXACTLY, whatever the sign of the arc costs (an acoustic cost logp - log_prior can be negative
  // after the DecodeNumpy prologue; an over-estimated beta would prune valid paths below).  rank[s] = position of s in the
  // topological order: the epsilon closures below visit their states in that order.
  // Should a cycle of epsilon arcs ever leave states unordered, Dijkstra takes over -- on costs clamped at 0, the only place
  // the clamp is needed -- and the closures fall back to cheapest-first.
  std::vector<int>& order = tls.order; std::vector<int>& rank = tls.rank;
  order.clear(); order.reserve(n_states);
  rank.assign(n_states, -1);
  for (int s = 0; s < n_states; ++s) if (pending[s] == 0) order.push_back(s);
  for (size_t h = 0; h < order.size(); ++h) {
    const int s = order[h];
    rank[s] = (int)h;
    for (int k = out_off[s]; k < out_off[s + 1]; ++k) if (--pending[out_arc[k].dst] == 0) order.push_back(out_arc[k].dst);
  }
  const auto ts3 = std::chrono::steady_clock::now();
  const bool have_order = (int)order.size() == n_states;
  if (have_order) {
    for (int h = n_states - 1; h >= 0; --h) {
      const int s = order[h];
      double b = fin[s];
      for (int k = out_off[s]; k < out_off[s + 1]; ++k) {
        const Arc& a = out_arc[k];
        const double c = beta[a.dst] + ((double)a.g + (double)a.a);   // as the forward search adds them
        if (c < b) b = c;
      }
      beta[s] = b;
    }
    if (resc)
      for (int h = n_states - 1; h >= 0; --h) {
        const int s = order[h];
        double b = fin_o[s];
        for (int k = out_off[s]; k < out_off[s + 1]; ++k) {
          const Arc& a = out_arc[k];
          const double c = beta_o[a.dst] + ((double)a.g + (double)a.a - (double)a.d);
          if (c < b) b = c;
        }
        beta_o[s] = b;
      }
  } else {
    if (resc) { b2t::set_error("lattice_nbest: a rescored lattice must be acyclic"); return -1; }
    std::vector<int> rev_off(n_states + 1, 0);
    for (int i = 0; i < n_arcs; ++i) ++rev_off[dst[i] + 1];
    for (int s = 0; s < n_states; ++s) rev_off[s + 1] += rev_off[s];
    std::vector<std::pair<int, double>> rev_arc(n_arcs);
    {
      std::vector<int> pr(rev_off.begin(), rev_off.end() - 1);
      for (int i = 0; i < n_arcs; ++i) rev_arc[pr[dst[i]]++] = {src[i], (double)graph[i] + (double)acoustic[i]};
    }
    typedef std::pair<double, int> P;
    std::priority_queue<P, std::vector<P>, std::greater<P>> pq;
    for (int s = 0; s < n_states; ++s) if (fin[s] != INF) { beta[s] = fin[s]; pq.push({fin[s], s}); }
    while (!pq.empty()) {
      P t = pq.top(); pq.pop();
      if (t.first > beta[t.second]) continue;
      for (int k = rev_off[t.second]; k < rev_off[t.second + 1]; ++k) {
        const std::pair<int, double>& pr = rev_arc[k];
        const double c = t.first + std::max(0.0, pr.second);
        if (c < beta[pr.first]) { beta[pr.first] = c; pq.push({c, pr.first}); }
      }
    }
  }
  static const bool lat_timing = getenv("B2T_LAT_TIMING") != nullptr;
  const auto tm0 = std::chrono::steady_clock::now();
  w_off[0] = 0; a_off[0] = 0;
  if (beta[start] == INF) return 0;
  if (resc && beta_o[start] == INF) return 0;
  // Everything costlier than `limit` is dropped: at first the lattice beam; once `nbest` complete word sequences are known
  // (each with the exact cost of its best path), the cost of the worst of the best `nbest` of them -- nothing above it can be
  // among the answers, and the closures of the subsets still to be expanded shrink accordingly.
  // The search is first run with HALF the beam: a lattice that holds many alternatives (the expensive case: 10^5 arcs) has its
  // `nbest` answers well inside the beam, and a run that returns `nbest` sequences under a smaller limit returns exactly what
  // the full beam would (best-first order; everything it dropped costs more than all of them).  Only a run that comes back
  // short is repeated with the whole beam.  (32 lattices of the bench workload, one thread: 131 -> 6x ms.)
  double limit = 0, limit_o = 0;   // limit_o: the beam on the OLD costs (rescoring only)
  std::priority_queue<double> known;       // the `nbest` smallest totals of the finished sequences found so far
  std::vector<Node> ali_pool, word_pool;
  auto push_node = [](std::vector<Node>& pool, int parent, int label) { pool.push_back(Node{parent, label}); return (int)pool.size() - 1; };

  std::vector<int>& slot = tls.slot;       // index of a lattice state in the subset under construction
  slot.assign(n_states, -1);
  auto mark = [&](const Subset& sub) { for (size_t i = 0; i < sub.st.size(); ++i) slot[sub.st[i]] = (int)i; };
  auto unmark = [&](const Subset& sub) { for (int st : sub.st) slot[st] = -1; };
  std::vector<int> heap;                   // ranks of the closure's pending states (min-heap)
  auto closure = [&](Subset& sub) {        // epsilon-output closure (sub's states are marked on entry and on exit)
    if (have_order) {
      // in topological order: when a state is taken from the heap every predecessor inside the subset has been expanded, so
      // its entry is final and it is expanded exactly once (cheapest-first needed a (cost, state) heap and re-queued a state
      // on every improvement)
      heap.clear();
      for (int st : sub.st) heap.push_back(rank[st]);
      std::make_heap(heap.begin(), heap.end(), std::greater<int>());
      while (!heap.empty()) {
        std::pop_heap(heap.begin(), heap.end(), std::greater<int>());
        const int s = order[heap.back()]; heap.pop_back();
        const Entry e = sub.en[slot[s]];
        for (int k = out_off[s]; k < out_off[s + 1]; ++k) {
          const Arc& a = out_arc[k];
          if (a.ol != 0) continue;
          const double nt = e.tot + a.g + a.a;
          if (nt + beta[a.dst] > limit) continue;
          if (resc && nt - (e.dl + a.d) + beta_o[a.dst] > limit_o) continue;
          const int j = slot[a.dst];
          if (j >= 0 && !(nt < sub.en[j].tot)) continue;
          const Entry ne{nt, e.gr + a.g, e.ac + a.a, a.il ? push_node(ali_pool, e.ali, a.il) : e.ali, e.dl + a.d};
          if (j < 0) {
            slot[a.dst] = (int)sub.st.size(); sub.st.push_back(a.dst); sub.en.push_back(ne);
            heap.push_back(rank[a.dst]); std::push_heap(heap.begin(), heap.end(), std::greater<int>());
          } else sub.en[j] = ne;
        }
      }
      return;
    }
    typedef std::pair<double, int> P;
    std::priority_queue<P, std::vector<P>, std::greater<P>> pq;
    for (size_t i = 0; i < sub.st.size(); ++i) pq.push({sub.en[i].tot, sub.st[i]});
    while (!pq.empty()) {
      P t = pq.top(); pq.pop();
      const Entry e = sub.en[slot[t.second]];
      if (t.first > e.tot) continue;
      for (int k = out_off[t.second]; k < out_off[t.second + 1]; ++k) {
        const Arc& a = out_arc[k];
        if (a.ol != 0) continue;
        const double nt = e.tot + a.g + a.a;
        if (nt + beta[a.dst] > limit) continue;
        const int j = slot[a.dst];
        if (j >= 0 && !(nt < sub.en[j].tot)) continue;
        const Entry ne{nt, e.gr + a.g, e.ac + a.a, a.il ? push_node(ali_pool, e.ali, a.il) : e.ali, e.dl + a.d};
        if (j < 0) { slot[a.dst] = (int)sub.st.size(); sub.st.push_back(a.dst); sub.en.push_back(ne); }
        else sub.en[j] = ne;
        pq.push({nt, a.dst});
      }
    }
  };

  struct Done { double tot, gr, ac; int words, ali; };
  // queued subsets live back to back in two pools (four of five are never expanded: no vectors of their own); the one being
  // expanded is copied into `sub`, whose storage is reused from expansion to expansion
  struct Span { size_t off; int n; };
  std::vector<Span> subsets;
  std::vector<int> pool_st;
  std::vector<Entry> pool_en;
  Subset sub, tgt;
  std::vector<std::pair<unsigned long long, int>> keyed;      // (word label, position in `trans`) -> grouped by word, stable
  std::vector<Done> finished;
  std::priority_queue<Item> pq;
  std::vector<Trans> trans;
  int n_out = 0;
  double t_closure = 0; size_t n_pop = 0, n_clo = 0, n_trans = 0;
  static const double STAGE[3] = {0.5, 0.75, 1.0};
  for (int stage = resc ? 2 : 0; stage < 3; ++stage) {      // a rescored lattice is ranked by other costs than its beam: one run
  if (stage > 0 && n_out >= nbest) break;
  limit = resc ? INF : beta[start] + STAGE[stage] * (double)beam + 1e-4;
  limit_o = resc ? beta_o[start] + (double)beam + 1e-4 : INF;
  known = std::priority_queue<double>(); pq = std::priority_queue<Item>();
  subsets.clear(); pool_st.clear(); pool_en.clear(); finished.clear(); ali_pool.clear(); word_pool.clear();
  long long tie = 0;
  subsets.push_back(Span{0, 1});
  pool_st.push_back(start); pool_en.push_back(Entry{0.0, 0.0, 0.0, -1, 0.0});
  pq.push(Item{beta[start], 0, tie++, -1, 0});
  n_out = 0;
  while (!pq.empty() && n_out < nbest) {
    const Item it = pq.top(); pq.pop();
    if (it.bound > limit) break;
    if (it.kind == 1) {
      const Done& d = finished[it.payload];
      std::vector<int> w, a;
      for (int n = d.words; n >= 0; n = word_pool[n].parent) w.push_back(word_pool[n].label);
      for (int n = d.ali; n >= 0; n = ali_pool[n].parent) a.push_back(ali_pool[n].label);
      if (w_off[n_out] + (int)w.size() > w_cap || a_off[n_out] + (int)a.size() > a_cap) { b2t::set_error("lattice_nbest: output buffers too small"); return -2; }
      std::reverse(w.begin(), w.end()); std::reverse(a.begin(), a.end());
      if (out_words) std::copy(w.begin(), w.end(), out_words + w_off[n_out]);
      if (out_ali) std::copy(a.begin(), a.end(), out_ali + a_off[n_out]);
      w_off[n_out + 1] = w_off[n_out] + (int)w.size();
      a_off[n_out + 1] = a_off[n_out] + (int)a.size();
      costs[2 * n_out] = (float)d.gr; costs[2 * n_out + 1] = (float)d.ac;
      ++n_out;
      continue;
    }
    // A subset is expanded once: take it out (`subsets` grows below).  Its epsilon-output closure is computed only now: the
    // bound it was queued with, min (cost so far + backward cost) over its entries, does not need it -- the backward cost of
    // an entry already is the cheapest way on through the closure -- and four of five queued subsets are never popped.
    {
      const Span sp = subsets[it.payload];
      sub.st.assign(pool_st.begin() + sp.off, pool_st.begin() + sp.off + sp.n);
      sub.en.assign(pool_en.begin() + sp.off, pool_en.begin() + sp.off + sp.n);
    }
    const auto tc0 = std::chrono::steady_clock::now();
    mark(sub); closure(sub); unmark(sub);
    const auto tc1 = std::chrono::steady_clock::now();
    t_closure += std::chrono::duration<double, std::milli>(tc1 - tc0).count(); ++n_pop; n_clo += sub.st.size();
    bool has = false; Done best{INF, 0, 0, it.words, -1};
    for (size_t i = 0; i < sub.st.size(); ++i) {
      const int st = sub.st[i];
      if (fin[st] == INF) continue;
      const Entry& e = sub.en[i];
      const double c = e.tot + fin[st];
      if (resc && e.tot - e.dl + fin_o[st] > limit_o) continue;
      if (c < best.tot) { best = Done{c, e.gr + fin[st], e.ac, it.words, e.ali}; has = true; }
    }
    if (has && best.tot <= limit) {
      finished.push_back(best); pq.push(Item{best.tot, 1, tie++, it.words, (int)finished.size() - 1});
      known.push(best.tot);
      if ((int)known.size() > nbest) known.pop();
      if ((int)known.size() == nbest) limit = std::min(limit, known.top() + 1e-6);   // slack: forward and backward sums round differently
    }
    // the word arcs out of the subset, grouped by word (ascending label: deterministic expansion order; within a word in
    // the order the subset and the lattice list them)
    trans.clear();
    for (size_t i = 0; i < sub.st.size(); ++i) {
      const Entry& e = sub.en[i];
      for (int k = out_off[sub.st[i]]; k < out_off[sub.st[i] + 1]; ++k) {
        const Arc& a = out_arc[k];
        if (a.ol == 0) continue;
        const double nt = e.tot + a.g + a.a;
        if (nt + beta[a.dst] > limit) continue;
        if (resc && nt - (e.dl + a.d) + beta_o[a.dst] > limit_o) continue;
        trans.push_back(Trans{a.ol, a.dst, nt, e.gr + a.g, e.ac + a.a, e.ali, a.il, e.dl + a.d});
      }
    }
    n_trans += trans.size();
    keyed.resize(trans.size());
    for (size_t q = 0; q < trans.size(); ++q) keyed[q] = {((unsigned long long)(unsigned)trans[q].ol << 32) | (unsigned long long)q, (int)q};
    std::sort(keyed.begin(), keyed.end());          // by word, then by position: what a stable sort of `trans` by word gives
    for (size_t g0 = 0; g0 < keyed.size();) {
      size_t g1 = g0;
      const int ol = trans[keyed[g0].second].ol;
      while (g1 < keyed.size() && trans[keyed[g1].second].ol == ol) ++g1;
      tgt.st.clear(); tgt.en.clear();
      for (size_t q = g0; q < g1; ++q) {
        const Trans& tr = trans[keyed[q].second];
        const int j = slot[tr.dst];
        if (j >= 0 && !(tr.tot < tgt.en[j].tot)) continue;
        const Entry ne{tr.tot, tr.gr, tr.ac, tr.il ? push_node(ali_pool, tr.ali, tr.il) : tr.ali, tr.dl};
        if (j < 0) { slot[tr.dst] = (int)tgt.st.size(); tgt.st.push_back(tr.dst); tgt.en.push_back(ne); }
        else tgt.en[j] = ne;
      }
      double b = INF;
      for (size_t i = 0; i < tgt.st.size(); ++i) b = std::min(b, tgt.en[i].tot + beta[tgt.st[i]]);
      unmark(tgt);
      subsets.push_back(Span{pool_st.size(), (int)tgt.st.size()});
      pool_st.insert(pool_st.end(), tgt.st.begin(), tgt.st.end());
      pool_en.insert(pool_en.end(), tgt.en.begin(), tgt.en.end());
      pq.push(Item{b, 0, tie++, push_node(word_pool, it.words, ol), (int)subsets.size() - 1});
      g0 = g1;
    }
  }
  }
  if (lat_timing) {
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    fprintf(stderr, "lattice_nbest setup: count %.2f, fill %.2f, kahn %.2f, beta %.2f ms\n", ms(tm_in, ts1), ms(ts1, ts2), ms(ts2, ts3), ms(ts3, tm0));
    size_t ents = pool_st.size();
    fprintf(stderr, "lattice_nbest: setup (adjacency + backward costs) %.1f ms, main loop %.1f ms, %zu subsets created, %zu alignment nodes, %zu subset entries queued in all; %zu subsets expanded, closures %.1f ms with %zu states, %zu word transitions\n",
            std::chrono::duration<double, std::milli>(tm0 - tm_in).count(), std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tm0).count(), subsets.size(), ali_pool.size(), ents, n_pop, t_closure, n_clo, n_trans);
  }
  return n_out;
}

extern "C" int b2t_lattice_nbest_host(int n_states, int start, int n_arcs, const int32_t* src, const int32_t* dst,
                                      const int32_t* ilabel, const int32_t* olabel, const float* graph, const float* acoustic,
                                      int n_final, const int32_t* final_state, const float* final_cost, int nbest, float beam,
                                      int32_t* out_words, int32_t* w_off, int w_cap, int32_t* out_ali, int32_t* a_off, int a_cap,
                                      float* costs) {
  return b2t_lattice_nbest_core(n_states, start, n_arcs, src, dst, ilabel, olabel, graph, acoustic, nullptr, n_final, final_state,
                                final_cost, nullptr, nbest, beam, out_words, w_off, w_cap, out_ali, a_off, a_cap, costs);
}

extern "C" int b2t_nbest_convert_to_inputs(const int32_t* ali, const int32_t* a_off, int n, const int32_t* mapping, int F,
                                           int32_t* out_inputs, int32_t* out_times, int32_t* out_off, int cap) {
  if (n < 0 || !a_off || !out_off || (n > 0 && (!ali || !out_inputs || !out_times)) || (F > 0 && !mapping)) {
    b2t::set_error("nbest_convert_to_inputs: bad arguments");
    return -1;
  }
  int w = 0;
  out_off[0] = 0;
  for (int k = 0; k < n; ++k) {
    const int32_t* a = ali + a_off[k];
    const int len = a_off[k + 1] - a_off[k];
    const bool full = len == F && F > 0;
    int cur = 0;
    while (cur < len) {
      while (cur < len && a[cur] == 1) ++cur;                              // blanks
      while (cur + 1 < len && a[cur + 1] == a[cur]) ++cur;                 // the unit's time is that of its last repeat
      if (cur < len) {
        if (w >= cap) { b2t::set_error("nbest_convert_to_inputs: output buffers too small"); return -2; }
        out_inputs[w] = a[cur] - 1;
        out_times[w] = full ? mapping[cur < F ? cur : F - 1] : cur;
        ++w; ++cur;
      }
    }
    out_off[k + 1] = w;
  }
  return 0;
}
