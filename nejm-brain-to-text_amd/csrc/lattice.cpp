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

struct Arc { int il, ol, dst; float g, a; };
struct Entry { double tot, gr, ac; int ali; };          // ali: node of the alignment list (-1 = empty)
struct Node { int parent, label; };
struct Subset { std::vector<int> st; std::vector<Entry> en; };   // lattice states and their best entries, insertion order
struct Trans { int ol, dst; double tot, gr, ac; int ali, il; };   // a word arc out of a subset

struct Item {
  double bound; int kind; long long tie; int words; int payload;   // kind 0: subset (payload = index), 1: finished (payload = index)
  bool operator<(const Item& o) const {                             // priority_queue is a max-heap: invert
    if (bound != o.bound) return bound > o.bound;
    if (kind != o.kind) return kind < o.kind;                       // finished sequences before subsets of equal bound
    return tie > o.tie;
  }
};

}  // namespace

extern "C" int b2t_lattice_nbest_host(int n_states, int start, int n_arcs, const int32_t* src, const int32_t* dst,
                                      const int32_t* ilabel, const int32_t* olabel, const float* graph, const float* acoustic,
                                      int n_final, const int32_t* final_state, const float* final_cost, int nbest, float beam,
                                      int32_t* out_words, int32_t* w_off, int w_cap, int32_t* out_ali, int32_t* a_off, int a_cap,
                                      float* costs) {
  if (n_states <= 0 || start < 0 || start >= n_states || n_arcs < 0 || nbest <= 0 || !w_off || !a_off || !costs) {
    b2t::set_error("lattice_nbest: bad arguments");
    return -1;
  }
  const double INF = INFINITY;
  const auto tm_in = std::chrono::steady_clock::now();
  // CSR adjacency both ways (lattices reach 10^5 arcs: no per-state vectors)
  std::vector<int> out_off(n_states + 1, 0), rev_off(n_states + 1, 0);
  for (int i = 0; i < n_arcs; ++i) {
    if (src[i] < 0 || src[i] >= n_states || dst[i] < 0 || dst[i] >= n_states) { b2t::set_error("lattice_nbest: arc %d out of range", i); return -1; }
    ++out_off[src[i] + 1]; ++rev_off[dst[i] + 1];
  }
  for (int s = 0; s < n_states; ++s) { out_off[s + 1] += out_off[s]; rev_off[s + 1] += rev_off[s]; }
  std::vector<Arc> out_arc(n_arcs);
  std::vector<std::pair<int, double>> rev_arc(n_arcs);
  {
    std::vector<int> po(out_off.begin(), out_off.end() - 1), pr(rev_off.begin(), rev_off.end() - 1);
    for (int i = 0; i < n_arcs; ++i) {                       // arc order within a state = input order (deterministic ties)
      out_arc[po[src[i]]++] = Arc{ilabel[i], olabel[i], dst[i], graph[i], acoustic[i]};
      rev_arc[pr[dst[i]]++] = {src[i], (double)graph[i] + (double)acoustic[i]};   // as the forward search adds them
    }
  }
  std::vector<double> fin(n_states, INF), beta(n_states, INF);
  for (int i = 0; i < n_final; ++i) fin[final_state[i]] = std::min(fin[final_state[i]], (double)final_cost[i]);
  // beta: cheapest completion incl. the final cost.  The lattice is acyclic, so one relaxation sweep in reverse
  // topological order (Kahn, O(states + arcs)) gives it EXACTLY, whatever the sign of the arc costs (an acoustic cost
  // logp - log_prior can be negative after the DecodeNumpy prologue; an over-estimated beta would prune valid paths below).
  // Should a cycle of epsilon arcs ever leave states unordered, Dijkstra takes over -- on costs clamped at 0, the only
  // place the clamp is needed.
  bool have_beta = false;
  {
    std::vector<int> pending(n_states);
    for (int s = 0; s < n_states; ++s) pending[s] = out_off[s + 1] - out_off[s];
    std::vector<int> order; order.reserve(n_states);
    for (int s = 0; s < n_states; ++s) if (pending[s] == 0) order.push_back(s);
    for (int s = 0; s < n_states; ++s) beta[s] = fin[s];
    for (size_t h = 0; h < order.size(); ++h) {
      const int s = order[h];
      for (int k = rev_off[s]; k < rev_off[s + 1]; ++k) {
        const std::pair<int, double>& pr = rev_arc[k];
        const double c = beta[s] + pr.second;
        if (c < beta[pr.first]) beta[pr.first] = c;
        if (--pending[pr.first] == 0) order.push_back(pr.first);
      }
    }
    have_beta = (int)order.size() == n_states;
    if (!have_beta) std::fill(beta.begin(), beta.end(), INF);
  }
  if (!have_beta) {
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
  const double limit = beta[start] + beam + 1e-4;
  std::vector<Node> ali_pool, word_pool;
  auto push_node = [](std::vector<Node>& pool, int parent, int label) { pool.push_back(Node{parent, label}); return (int)pool.size() - 1; };

  std::vector<int> slot(n_states, -1);     // index of a lattice state in the subset under construction
  auto mark = [&](const Subset& sub) { for (size_t i = 0; i < sub.st.size(); ++i) slot[sub.st[i]] = (int)i; };
  auto unmark = [&](const Subset& sub) { for (int st : sub.st) slot[st] = -1; };
  auto closure = [&](Subset& sub) {        // epsilon-output closure, cheapest first (sub's states are marked on entry and on exit)
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
        const Entry ne{nt, e.gr + a.g, e.ac + a.a, a.il ? push_node(ali_pool, e.ali, a.il) : e.ali};
        if (j < 0) { slot[a.dst] = (int)sub.st.size(); sub.st.push_back(a.dst); sub.en.push_back(ne); }
        else sub.en[j] = ne;
        pq.push({nt, a.dst});
      }
    }
  };

  struct Done { double tot, gr, ac; int words, ali; };
  std::vector<Subset> subsets;
  std::vector<Done> finished;
  std::priority_queue<Item> pq;
  std::vector<Trans> trans;
  long long tie = 0;
  subsets.emplace_back();
  subsets[0].st.push_back(start); subsets[0].en.push_back(Entry{0.0, 0.0, 0.0, -1});
  pq.push(Item{beta[start], 0, tie++, -1, 0});
  int n_out = 0;
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
    Subset sub = std::move(subsets[it.payload]);
    mark(sub); closure(sub); unmark(sub);
    bool has = false; Done best{INF, 0, 0, it.words, -1};
    for (size_t i = 0; i < sub.st.size(); ++i) {
      const int st = sub.st[i];
      if (fin[st] == INF) continue;
      const Entry& e = sub.en[i];
      const double c = e.tot + fin[st];
      if (c < best.tot) { best = Done{c, e.gr + fin[st], e.ac, it.words, e.ali}; has = true; }
    }
    if (has && best.tot <= limit) { finished.push_back(best); pq.push(Item{best.tot, 1, tie++, it.words, (int)finished.size() - 1}); }
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
        trans.push_back(Trans{a.ol, a.dst, nt, e.gr + a.g, e.ac + a.a, e.ali, a.il});
      }
    }
    std::stable_sort(trans.begin(), trans.end(), [](const Trans& x, const Trans& y) { return x.ol < y.ol; });
    for (size_t g0 = 0; g0 < trans.size();) {
      size_t g1 = g0;
      while (g1 < trans.size() && trans[g1].ol == trans[g0].ol) ++g1;
      Subset tgt;
      for (size_t q = g0; q < g1; ++q) {
        const Trans& tr = trans[q];
        const int j = slot[tr.dst];
        if (j >= 0 && !(tr.tot < tgt.en[j].tot)) continue;
        const Entry ne{tr.tot, tr.gr, tr.ac, tr.il ? push_node(ali_pool, tr.ali, tr.il) : tr.ali};
        if (j < 0) { slot[tr.dst] = (int)tgt.st.size(); tgt.st.push_back(tr.dst); tgt.en.push_back(ne); }
        else tgt.en[j] = ne;
      }
      double b = INF;
      for (size_t i = 0; i < tgt.st.size(); ++i) b = std::min(b, tgt.en[i].tot + beta[tgt.st[i]]);
      unmark(tgt);
      const int ol = trans[g0].ol;
      subsets.push_back(std::move(tgt));
      pq.push(Item{b, 0, tie++, push_node(word_pool, it.words, ol), (int)subsets.size() - 1});
      g0 = g1;
    }
  }
  if (lat_timing) {
    size_t ents = 0; for (const Subset& x : subsets) ents += x.st.size();
    fprintf(stderr, "lattice_nbest: setup (adjacency + backward costs) %.1f ms, main loop %.1f ms, %zu subsets created, %zu alignment nodes, %zu entries left in unexpanded subsets\n",
            std::chrono::duration<double, std::milli>(tm0 - tm_in).count(), std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tm0).count(), subsets.size(), ali_pool.size(), ents);
  }
  return n_out;
}
