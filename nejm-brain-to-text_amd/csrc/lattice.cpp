// lattice.cpp — HOST side of FinalizeSearch (ctc_wfst_beam_search.cc:123-160): the n-best distinct word sequences of the
// pruned token lattice the GPU search leaves behind.  The reference gets them from DeterminizeLatticePruned (one path per
// word sequence: the best one; lattice-faster-decoder.cc:193-213) followed by fst::ShortestPath(nbest); this is that
// definition computed directly: subset construction over the word labels (a determinised state is a set of lattice states
// with their best cost so far), expanded best-first with the exact backward cost as the bound, so that sequences come
// out in order of total cost and the search stops after `nbest` of them or at `beam` above the best.
// Runs once per utterance on a lattice of a few thousand arcs; the per-frame hot loop is csrc/wfst.hip.
#include <algorithm>
#include <queue>
#include <unordered_map>
#include <vector>
#include <cmath>
#include "common.h"

namespace {

struct Arc { int il, ol, dst; float g, a; };
struct Entry { double tot, gr, ac; int ali; };          // ali: node of the alignment list (-1 = empty)
struct Node { int parent, label; };
typedef std::unordered_map<int, Entry> Subset;

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
      rev_arc[pr[dst[i]]++] = {src[i], std::max(0.0, (double)graph[i] + (double)acoustic[i])};
    }
  }
  std::vector<double> fin(n_states, INF), beta(n_states, INF);
  for (int i = 0; i < n_final; ++i) fin[final_state[i]] = std::min(fin[final_state[i]], (double)final_cost[i]);
  {   // beta: cheapest completion incl. the final cost (arc costs are >= 0 once the per-frame offsets are taken out)
    typedef std::pair<double, int> P;
    std::priority_queue<P, std::vector<P>, std::greater<P>> pq;
    for (int s = 0; s < n_states; ++s) if (fin[s] != INF) { beta[s] = fin[s]; pq.push({fin[s], s}); }
    while (!pq.empty()) {
      P t = pq.top(); pq.pop();
      if (t.first > beta[t.second]) continue;
      for (int k = rev_off[t.second]; k < rev_off[t.second + 1]; ++k) {
        const std::pair<int, double>& pr = rev_arc[k];
        const double c = t.first + pr.second;
        if (c < beta[pr.first]) { beta[pr.first] = c; pq.push({c, pr.first}); }
      }
    }
  }
  w_off[0] = 0; a_off[0] = 0;
  if (beta[start] == INF) return 0;
  const double limit = beta[start] + beam + 1e-4;
  std::vector<Node> ali_pool, word_pool;
  auto push_node = [](std::vector<Node>& pool, int parent, int label) { pool.push_back(Node{parent, label}); return (int)pool.size() - 1; };

  auto closure = [&](Subset& sub) {
    typedef std::pair<double, int> P;
    std::priority_queue<P, std::vector<P>, std::greater<P>> pq;
    for (auto& kv : sub) pq.push({kv.second.tot, kv.first});
    while (!pq.empty()) {
      P t = pq.top(); pq.pop();
      const Entry e = sub[t.second];
      if (t.first > e.tot) continue;
      for (int k = out_off[t.second]; k < out_off[t.second + 1]; ++k) {
        const Arc& a = out_arc[k];
        if (a.ol != 0) continue;
        const double nt = e.tot + a.g + a.a;
        if (nt + beta[a.dst] > limit) continue;
        auto it = sub.find(a.dst);
        if (it == sub.end() || nt < it->second.tot) {
          sub[a.dst] = Entry{nt, e.gr + a.g, e.ac + a.a, a.il ? push_node(ali_pool, e.ali, a.il) : e.ali};
          pq.push({nt, a.dst});
        }
      }
    }
  };

  struct Done { double tot, gr, ac; int words, ali; };
  std::vector<Subset> subsets;
  std::vector<Done> finished;
  std::priority_queue<Item> pq;
  long long tie = 0;
  subsets.emplace_back();
  subsets[0][start] = Entry{0.0, 0.0, 0.0, -1};
  closure(subsets[0]);
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
    const Subset sub = subsets[it.payload];   // (copy: `subsets` grows below)
    bool has = false; Done best{INF, 0, 0, it.words, -1};
    for (auto& kv : sub) {
      if (fin[kv.first] == INF) continue;
      const double c = kv.second.tot + fin[kv.first];
      if (c < best.tot) { best = Done{c, kv.second.gr + fin[kv.first], kv.second.ac, it.words, kv.second.ali}; has = true; }
    }
    if (has && best.tot <= limit) { finished.push_back(best); pq.push(Item{best.tot, 1, tie++, it.words, (int)finished.size() - 1}); }
    std::unordered_map<int, Subset> by_word;
    for (auto& kv : sub) {
      for (int k = out_off[kv.first]; k < out_off[kv.first + 1]; ++k) {
        const Arc& a = out_arc[k];
        if (a.ol == 0) continue;
        const double nt = kv.second.tot + a.g + a.a;
        if (nt + beta[a.dst] > limit) continue;
        Subset& tgt = by_word[a.ol];
        auto f = tgt.find(a.dst);
        if (f == tgt.end() || nt < f->second.tot)
          tgt[a.dst] = Entry{nt, kv.second.gr + a.g, kv.second.ac + a.a, a.il ? push_node(ali_pool, kv.second.ali, a.il) : kv.second.ali};
      }
    }
    std::vector<int> labels;
    for (auto& kv : by_word) labels.push_back(kv.first);
    std::sort(labels.begin(), labels.end());          // deterministic expansion order
    for (int ol : labels) {
      Subset& tgt = by_word[ol];
      closure(tgt);
      double b = INF;
      for (auto& kv : tgt) b = std::min(b, kv.second.tot + beta[kv.first]);
      subsets.push_back(tgt);
      pq.push(Item{b, 0, tie++, push_node(word_pool, it.words, ol), (int)subsets.size() - 1});
    }
  }
  return n_out;
}
