// graphc.cpp — host-side decode-graph compiler: the FST algebra the reference's recipe runs through OpenFST /
// Kaldi binaries (language_model/tools/fst/make_tlg.sh:29-46: fsttablecompose | fstdeterminizestar --use-log=true |
// fstminimizeencoded | fstarcsort, then fsttablecompose with T) on flat arrays, behind a C ABI (include/b2t.h, b2t_fst_*).
// Written for graphs of the reference's size class (125 k words, 10^6-10^8 arcs): CSR storage, hashed state tuples, no
// per-arc heap objects.  Semantics restated from:
//   composition            epsilon-matching filter of fst::Compose (filter states 0 / 1 / 2), as nejm-brain-to-text_amd/wfst.py
//   determinize-star       language_model/runtime/core/kaldi/fstext/determinize-star-inl.h:184-1130 (subsets of (state, output
//                          string, residual weight); epsilon closure; common output prefix and total weight divided out of a
//                          transition; output strings expanded into chains of input-epsilon arcs), tropical or log semiring
//   minimize-encoded       language_model/runtime/core/kaldi/fstext/fstext-utils.h:110-116 (quantise weights, encode (ilabel,
//                          olabel, weight) as one label, minimise the deterministic acceptor, decode)
// OpenFST itself is not in the image and not in the reference checkout: nothing here can be compared with its output
// files; tests/test_graphc.py checks the definitions instead (same weighted relation before and after, determinism,
// minimality against a brute-force Myhill-Nerode partition, equality with the Python composition).
#include <condition_variable>
#include <mutex>
#include <atomic>
#include <system_error>
#include <thread>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <array>
#include <chrono>
#include <stdlib.h>
#include <deque>
#include <functional>
#include <queue>
#include <string>
#include <unordered_map>
#include <vector>
#include "common.h"

namespace b2t {
namespace {

struct HArc { int il, ol; float w; int nx; };
struct HFst {
  int start = -1;
  std::vector<int64_t> row;      // [n + 1]
  std::vector<HArc> arcs;
  std::vector<float> fin;        // [n], +inf = not final
  int n() const { return (int)fin.size(); }
};

const float FINF = INFINITY;

HFst* from_lists(int n, int start, const std::vector<int>& src, std::vector<HArc>& arcs, std::vector<float>& fin) {
  HFst* f = new HFst;
  f->start = start;
  f->fin.swap(fin);
  f->row.assign((size_t)n + 1, 0);
  for (int s : src) ++f->row[(size_t)s + 1];
  for (int s = 0; s < n; ++s) f->row[(size_t)s + 1] += f->row[s];
  f->arcs.resize(arcs.size());
  std::vector<int64_t> pos(f->row.begin(), f->row.end() - 1);
  for (size_t i = 0; i < arcs.size(); ++i) f->arcs[(size_t)pos[src[i]]++] = arcs[i];     // stable: arc order within a state kept
  return f;
}

inline float plus_w(float a, float b, bool log_sr) {
  if (!log_sr) return a < b ? a : b;
  if (a == FINF) return b;
  if (b == FINF) return a;
  const float m = a < b ? a : b, d = fabsf(a - b);
  return m - log1pf(expf(-d));
}

struct VecHash {
  size_t operator()(const std::vector<int>& v) const {
    size_t h = 1469598103934665603ull;
    for (int x : v) { h ^= (size_t)(unsigned)x; h *= 1099511628211ull; }
    return h;
  }
};

// ---- output-string repository (determinize-star-inl.h:42-180): id 0 = the empty string -------------------------------------
struct Strings {
  std::vector<std::vector<int>> seq{std::vector<int>()};
  std::unordered_map<std::vector<int>, int, VecHash> id{{std::vector<int>(), 0}};
  int of(const std::vector<int>& v) {
    auto it = id.find(v);
    if (it != id.end()) return it->second;
    seq.push_back(v);
    id.emplace(v, (int)seq.size() - 1);
    return (int)seq.size() - 1;
  }
  int append(int s, int label) { std::vector<int> v = seq[s]; v.push_back(label); return of(v); }
  int remove_prefix(int s, size_t k) { if (k == 0) return s; std::vector<int> v(seq[s].begin() + k, seq[s].end()); return of(v); }
};

struct Elem { int state, str; float w; };

struct SubsetHash {
  size_t operator()(const std::vector<Elem>* v) const {
    size_t h = 1469598103934665603ull;
    for (const Elem& e : *v) { h ^= (size_t)(unsigned)e.state * 0x9e3779b97f4a7c15ull + (size_t)(unsigned)e.str; h *= 1099511628211ull; }
    return h;
  }
};
struct SubsetEq {
  float delta;
  bool operator()(const std::vector<Elem>* a, const std::vector<Elem>* b) const {
    if (a->size() != b->size()) return false;
    for (size_t i = 0; i < a->size(); ++i) {
      const Elem &x = (*a)[i], &y = (*b)[i];
      if (x.state != y.state || x.str != y.str) return false;
      if (!(fabsf(x.w - y.w) <= delta) && !(x.w == y.w)) return false;
    }
    return true;
  }
};

}  // namespace
}  // namespace b2t

using namespace b2t;

#define FST(h) (reinterpret_cast<HFst*>(h))
#define CFST(h) (reinterpret_cast<const HFst*>(h))

extern "C" void* b2t_fst_from_arrays(int n_states, int start, long long n_arcs, const int32_t* src, const int32_t* il, const int32_t* ol,
                                     const float* w, const int32_t* dst, const float* final_cost) {
  if (n_states <= 0 || start < 0 || start >= n_states || n_arcs < 0 || !final_cost || (n_arcs > 0 && (!src || !il || !ol || !w || !dst))) {
    set_error("fst_from_arrays: bad arguments");
    return nullptr;
  }
  std::vector<int> s((size_t)n_arcs);
  std::vector<HArc> a((size_t)n_arcs);
  for (long long i = 0; i < n_arcs; ++i) {
    if (src[i] < 0 || src[i] >= n_states || dst[i] < 0 || dst[i] >= n_states) { set_error("fst_from_arrays: arc %lld out of range", i); return nullptr; }
    s[(size_t)i] = src[i]; a[(size_t)i] = HArc{il[i], ol[i], w[i], dst[i]};
  }
  std::vector<float> fin(final_cost, final_cost + n_states);
  return from_lists(n_states, start, s, a, fin);
}

extern "C" void b2t_fst_free(void* h) { delete FST(h); }

extern "C" int b2t_fst_info(const void* h, long long* out4) {
  B2T_REQUIRE(h && out4, "fst_info: null argument");
  const HFst* f = CFST(h);
  long long nf = 0;
  for (float c : f->fin) nf += c != FINF;
  out4[0] = f->n(); out4[1] = (long long)f->arcs.size(); out4[2] = f->start; out4[3] = nf;
  return 0;
}

extern "C" int b2t_fst_to_arrays(const void* h, long long* row, int32_t* il, int32_t* ol, float* w, int32_t* nx, float* final_cost) {
  B2T_REQUIRE(h && row && final_cost, "fst_to_arrays: null argument");
  const HFst* f = CFST(h);
  for (size_t i = 0; i < f->row.size(); ++i) row[i] = f->row[i];
  for (size_t i = 0; i < f->arcs.size(); ++i) { il[i] = f->arcs[i].il; ol[i] = f->arcs[i].ol; w[i] = f->arcs[i].w; nx[i] = f->arcs[i].nx; }
  for (size_t i = 0; i < f->fin.size(); ++i) final_cost[i] = f->fin[i];
  return 0;
}

// ---- arc sort (fstarcsort --sort_type=ilabel / olabel), stable -----------------------------------------------------------------
extern "C" void* b2t_fst_arcsort(const void* h, int by_olabel) {
  if (!h) { set_error("fst_arcsort: null"); return nullptr; }
  HFst* g = new HFst(*CFST(h));
  for (int s = 0; s < g->n(); ++s)
    std::stable_sort(g->arcs.begin() + g->row[s], g->arcs.begin() + g->row[(size_t)s + 1],
                     [&](const HArc& a, const HArc& b) { return by_olabel ? a.ol < b.ol : a.il < b.il; });
  return g;
}

// fst::ReadAndPrepareLmFst (kaldi/fstext/kaldi-fst-io.cc:129-147) without leaving C++: a grammar that is not an acceptor is
// projected on its OUTPUT labels, then arcs are sorted by input label.  *backoff_label: 0, unless the grammar is an acceptor in
// which no arc carries label 0 while arcs carry `disambig_id` (>= 0: words.txt's id of "#0" -- a grammar compiled with #0 on both
// sides).  (The Python host used to copy the whole grammar into numpy for the `ilabel != olabel` test: G_no_prune.fst has ~1e9 arcs.)
extern "C" void* b2t_fst_prepare_lm(const void* h, int disambig_id, int* backoff_label) {
  if (!h || !backoff_label) { set_error("fst_prepare_lm: null argument"); return nullptr; }
  const HFst& f = *CFST(h);
  bool acceptor = true, has0 = false, has_dis = false;
  for (const HArc& a : f.arcs) {
    acceptor = acceptor && a.il == a.ol;
    has0 = has0 || a.il == 0;
    has_dis = has_dis || (disambig_id >= 0 && a.il == disambig_id);
  }
  HFst* g = new HFst(f);
  *backoff_label = 0;
  if (!acceptor) { for (HArc& a : g->arcs) a.il = a.ol; }
  else if (!f.arcs.empty() && !has0 && has_dis) *backoff_label = disambig_id;
  for (int s = 0; s < g->n(); ++s)
    std::stable_sort(g->arcs.begin() + g->row[s], g->arcs.begin() + g->row[(size_t)s + 1], [](const HArc& a, const HArc& b) { return a.il < b.il; });
  return g;
}

// ---- composition with the epsilon-matching filter ---------------------------------------------------------------------------
extern "C" void* b2t_fst_compose(const void* ha, const void* hb) {
  if (!ha || !hb) { set_error("fst_compose: null"); return nullptr; }
  const HFst &a = *CFST(ha), &b = *CFST(hb);
  if ((uint64_t)a.n() >= (1ull << 30)) { set_error("fst_compose: the left operand has too many states"); return nullptr; }
  // b's arcs by (state, ilabel), original order kept among equal labels
  std::vector<HArc> bs(b.arcs);
  for (int s = 0; s < b.n(); ++s)
    std::stable_sort(bs.begin() + b.row[s], bs.begin() + b.row[(size_t)s + 1], [](const HArc& x, const HArc& y) { return x.il < y.il; });
  auto brange = [&](int s, int label, int64_t& lo, int64_t& hi) {
    const HArc* p0 = bs.data() + b.row[s];
    const HArc* p1 = bs.data() + b.row[(size_t)s + 1];
    lo = std::lower_bound(p0, p1, label, [](const HArc& x, int l) { return x.il < l; }) - bs.data();
    hi = std::upper_bound(p0, p1, label, [](int l, const HArc& x) { return l < x.il; }) - bs.data();
  };
  std::unordered_map<uint64_t, int> ids;
  ids.reserve(1 << 20);
  std::vector<uint64_t> keys;
  std::vector<int> src;
  std::vector<HArc> arcs;
  std::vector<float> fin;
  auto sid = [&](int sa, int sb, int fs) {
    const uint64_t k = ((uint64_t)sa << 34) | ((uint64_t)(uint32_t)sb << 2) | (uint64_t)fs;
    auto it = ids.find(k);
    if (it != ids.end()) return it->second;
    const int id = (int)keys.size();
    ids.emplace(k, id);
    keys.push_back(k);
    fin.push_back(FINF);
    return id;
  };
  if ((uint64_t)b.n() >= (1ull << 32)) { set_error("fst_compose: the right operand has too many states"); return nullptr; }
  // a's arcs by (state, olabel) with their position in the state's arc list: when the right-hand state has far fewer arcs
  // than the left-hand one (L's loop state carries two arcs per WORD, a state of G a handful), the labels are matched from
  // b's side -- what fsttablecompose's table matcher is for; 125 k words x 10^6 grammar states would otherwise be 10^11 probes.
  // The matches are then emitted in a's arc order, so the result (state numbering included) does not depend on the side.
  struct AO { int ol; int idx; };
  std::vector<AO> as(a.arcs.size());
  std::vector<int> a_nz((size_t)a.n(), 0);
  for (int s = 0; s < a.n(); ++s) {
    for (int64_t i = a.row[s]; i < a.row[(size_t)s + 1]; ++i) { as[(size_t)i] = AO{a.arcs[(size_t)i].ol, (int)(i - a.row[s])}; a_nz[(size_t)s] += a.arcs[(size_t)i].ol != 0; }
    std::stable_sort(as.begin() + a.row[s], as.begin() + a.row[(size_t)s + 1], [](const AO& x, const AO& y) { return x.ol < y.ol; });
  }
  struct Ev { int aidx; int64_t j; };          // j >= 0: a's arc aidx matched with b's (sorted) arc j; j = -1: a's arc has an epsilon output
  std::vector<Ev> ev;
  sid(a.start, b.start, 0);
  for (size_t q = 0; q < keys.size(); ++q) {
    const uint64_t k = keys[q];
    const int sa = (int)(k >> 34), sb = (int)((k >> 2) & 0xffffffffu), fs = (int)(k & 3);
    const int s = (int)q;
    if (a.fin[sa] != FINF && b.fin[sb] != FINF) fin[s] = a.fin[sa] + b.fin[sb];
    int64_t e0, e1;
    brange(sb, 0, e0, e1);
    const int64_t nb_nz = (b.row[(size_t)sb + 1] - b.row[sb]) - (e1 - e0);
    auto emit_match = [&](const HArc& x, const HArc& y) {
      const int d = sid(x.nx, y.nx, 0);
      src.push_back(s); arcs.push_back(HArc{x.il, y.ol, x.w + y.w, d});
    };
    auto emit_a_eps = [&](const HArc& x) {
      if (fs != 2) { const int d = sid(x.nx, sb, 1); src.push_back(s); arcs.push_back(HArc{x.il, 0, x.w, d}); }
      if (fs == 0)
        for (int64_t j = e0; j < e1; ++j) {
          const HArc& y = bs[(size_t)j];
          const int d = sid(x.nx, y.nx, 0);
          src.push_back(s); arcs.push_back(HArc{x.il, y.ol, x.w + y.w, d});
        }
    };
    if (nb_nz * 4 < (int64_t)a_nz[(size_t)sa]) {
      ev.clear();
      const AO* p0 = as.data() + a.row[sa];
      const AO* p1 = as.data() + a.row[(size_t)sa + 1];
      for (const AO* p = p0; p < p1 && p->ol == 0; ++p) ev.push_back(Ev{p->idx, -1});
      for (int64_t j = e1; j < b.row[(size_t)sb + 1]; ++j) {
        const int label = bs[(size_t)j].il;
        const AO* lo = std::lower_bound(p0, p1, label, [](const AO& x, int l) { return x.ol < l; });
        for (const AO* p = lo; p < p1 && p->ol == label; ++p) ev.push_back(Ev{p->idx, j});
      }
      std::sort(ev.begin(), ev.end(), [](const Ev& x, const Ev& y) { return x.aidx != y.aidx ? x.aidx < y.aidx : x.j < y.j; });
      for (const Ev& e : ev) {
        const HArc& x = a.arcs[(size_t)(a.row[sa] + e.aidx)];
        if (e.j < 0) emit_a_eps(x); else emit_match(x, bs[(size_t)e.j]);
      }
    } else {
      for (int64_t i = a.row[sa]; i < a.row[(size_t)sa + 1]; ++i) {
        const HArc& x = a.arcs[(size_t)i];
        if (x.ol != 0) {
          int64_t lo, hi;
          brange(sb, x.ol, lo, hi);
          for (int64_t j = lo; j < hi; ++j) emit_match(x, bs[(size_t)j]);
        } else {
          emit_a_eps(x);
        }
      }
    }
    if (fs != 1)
      for (int64_t j = e0; j < e1; ++j) {
        const HArc& y = bs[(size_t)j];
        const int d = sid(sa, y.nx, 2);
        src.push_back(s); arcs.push_back(HArc{0, y.ol, y.w, d});
      }
    if (keys.size() > (size_t)INT32_MAX - 8) { set_error("fst_compose: more than 2^31 states"); return nullptr; }
  }
  return from_lists((int)keys.size(), 0, src, arcs, fin);
}

// ---- trim (fstconnect): accessible and co-accessible states; start becomes 0, the others keep their order -------------------------
extern "C" void* b2t_fst_trim(const void* h) {
  if (!h) { set_error("fst_trim: null"); return nullptr; }
  const HFst& f = *CFST(h);
  const int n = f.n();
  std::vector<char> fw((size_t)n, 0), bw((size_t)n, 0);
  std::vector<int> st{f.start};
  fw[(size_t)f.start] = 1;
  while (!st.empty()) {
    const int s = st.back(); st.pop_back();
    for (int64_t i = f.row[s]; i < f.row[(size_t)s + 1]; ++i) { const int d = f.arcs[(size_t)i].nx; if (!fw[(size_t)d]) { fw[(size_t)d] = 1; st.push_back(d); } }
  }
  std::vector<int64_t> rrow((size_t)n + 1, 0);
  for (const HArc& a : f.arcs) ++rrow[(size_t)a.nx + 1];
  for (int s = 0; s < n; ++s) rrow[(size_t)s + 1] += rrow[s];
  std::vector<int> rsrc(f.arcs.size());
  {
    std::vector<int64_t> pos(rrow.begin(), rrow.end() - 1);
    for (int s = 0; s < n; ++s) for (int64_t i = f.row[s]; i < f.row[(size_t)s + 1]; ++i) rsrc[(size_t)pos[f.arcs[(size_t)i].nx]++] = s;
  }
  for (int s = 0; s < n; ++s) if (f.fin[s] != FINF) { bw[(size_t)s] = 1; st.push_back(s); }
  while (!st.empty()) {
    const int s = st.back(); st.pop_back();
    for (int64_t i = rrow[s]; i < rrow[(size_t)s + 1]; ++i) { const int p = rsrc[(size_t)i]; if (!bw[(size_t)p]) { bw[(size_t)p] = 1; st.push_back(p); } }
  }
  if (!(fw[(size_t)f.start] && bw[(size_t)f.start])) { set_error("fst_trim: the graph accepts nothing"); return nullptr; }
  std::vector<int> nid((size_t)n, -1);
  int m = 0;
  nid[(size_t)f.start] = m++;
  for (int s = 0; s < n; ++s) if (s != f.start && fw[(size_t)s] && bw[(size_t)s]) nid[(size_t)s] = m++;
  std::vector<int> src;
  std::vector<HArc> arcs;
  std::vector<float> fin((size_t)m, FINF);
  // arcs in the order of the ORIGINAL arc list of surviving states, sources renumbered (stable per state in from_lists)
  for (int s = 0; s < n; ++s) {
    if (nid[(size_t)s] < 0) continue;
    fin[(size_t)nid[(size_t)s]] = f.fin[s];
    for (int64_t i = f.row[s]; i < f.row[(size_t)s + 1]; ++i) {
      const HArc& a = f.arcs[(size_t)i];
      if (nid[(size_t)a.nx] >= 0) { src.push_back(nid[(size_t)s]); arcs.push_back(HArc{a.il, a.ol, a.w, nid[(size_t)a.nx]}); }
    }
  }
  return from_lists(m, 0, src, arcs, fin);
}

// ---- determinize-star ------------------------------------------------------------------------------------------------------------
extern "C" void* b2t_fst_determinize_star(const void* h, int use_log, float delta, long long max_states) {
  if (!h) { set_error("fst_determinize_star: null"); return nullptr; }
  const HFst& f = *CFST(h);
  const bool lg = use_log != 0;
  Strings rep;
  struct TArc { int il, str, nx; float w; };              // nx == -1: final weight
  std::vector<std::vector<TArc>> out;
  SubsetEq eq{delta};
  std::unordered_map<const std::vector<Elem>*, int, SubsetHash, SubsetEq> hash(1 << 16, SubsetHash(), eq);
  std::vector<std::vector<Elem>*> owned;
  std::deque<std::pair<std::vector<Elem>*, int>> Q;
  auto subset_id = [&](const std::vector<Elem>& sub) {
    auto it = hash.find(&sub);
    if (it != hash.end()) return it->second;
    std::vector<Elem>* keep = new std::vector<Elem>(sub);
    owned.push_back(keep);
    const int id = (int)out.size();
    hash.emplace(keep, id);
    out.emplace_back();
    Q.push_front({keep, id});                              // (allow_partial = false: determinize-star-inl.h:617-621)
    return id;
  };
  bool failed = false;
  std::string why;
  // epsilon closure of a subset: weights of all epsilon paths are combined with Plus; re-propagation while a state's
  // weight still moves by more than delta (determinize-star-inl.h:705-870)
  std::vector<int> idx_of((size_t)f.n(), -1);
  auto closure = [&](const std::vector<Elem>& in, std::vector<Elem>& outset) {
    struct Info { Elem e; float pending; bool queued; };
    std::vector<Info> info;
    std::deque<int> q;
    auto add = [&](int state, int str, float w) {
      int ix = idx_of[(size_t)state];
      if (ix < 0 || ix >= (int)info.size() || info[(size_t)ix].e.state != state) {
        idx_of[(size_t)state] = (int)info.size();
        info.push_back(Info{Elem{state, str, FINF}, w, true});
        q.push_back(state);
        return;
      }
      Info& I = info[(size_t)ix];
      if (I.e.str != str) { failed = true; why = "FST was not functional -> not determinizable (epsilon closure)"; return; }
      I.pending = plus_w(I.pending, w, lg);
      if (!I.queued) {
        const float tot = plus_w(I.e.w, I.pending, lg);
        if (!(fabsf(tot - I.e.w) <= delta)) { I.queued = true; q.push_back(state); }
      }
    };
    for (const Elem& e : in) add(e.state, e.str, e.w);
    long long guard = 0;
    while (!q.empty() && !failed) {
      const int s = q.front(); q.pop_front();
      Info& I = info[(size_t)idx_of[(size_t)s]];
      const float wproc = I.pending;
      I.e.w = plus_w(I.e.w, wproc, lg); I.pending = FINF; I.queued = false;
      const int str = I.e.str;
      for (int64_t i = f.row[s]; i < f.row[(size_t)s + 1]; ++i) {
        const HArc& a = f.arcs[(size_t)i];
        if (a.il != 0) continue;
        add(a.nx, a.ol == 0 ? str : rep.append(str, a.ol), wproc + a.w);
        if (failed) break;
      }
      if (++guard > 100000000ll) { failed = true; why = "epsilon closure did not converge (epsilon cycle in the log semiring?)"; }
    }
    outset.clear();
    for (Info& I : info) { if (I.pending != FINF) I.e.w = plus_w(I.e.w, I.pending, lg); outset.push_back(I.e); }
    std::sort(outset.begin(), outset.end(), [](const Elem& x, const Elem& y) { return x.state < y.state; });
    for (const Info& I : info) idx_of[(size_t)I.e.state] = -1;
  };
  {
    std::vector<Elem> init{Elem{f.start, 0, 0.f}};
    subset_id(init);
  }
  std::vector<Elem> closed, sub;
  std::vector<std::pair<int, Elem>> all;
  while (!Q.empty() && !failed) {
    auto pr = Q.front(); Q.pop_front();
    const int state = pr.second;
    closure(*pr.first, closed);
    if (failed) break;
    {  // ProcessFinal (:472-510)
      bool is_final = false; int fstr = 0; float fw = 0.f;
      for (const Elem& e : closed) {
        const float c = f.fin[(size_t)e.state];
        if (c == FINF) continue;
        if (!is_final) { fstr = e.str; fw = e.w + c; is_final = true; }
        else {
          if (fstr != e.str) { failed = true; why = "FST was not functional -> not determinizable (final weights)"; break; }
          fw = plus_w(fw, e.w + c, lg);
        }
      }
      if (is_final) out[(size_t)state].push_back(TArc{0, fstr, -1, fw});
    }
    if (failed) break;
    // ProcessTransitions (:545-600)
    all.clear();
    for (const Elem& e : closed)
      for (int64_t i = f.row[e.state]; i < f.row[(size_t)e.state + 1]; ++i) {
        const HArc& a = f.arcs[(size_t)i];
        if (a.il == 0) continue;
        all.push_back({a.il, Elem{a.nx, a.ol == 0 ? e.str : rep.append(e.str, a.ol), e.w + a.w}});
      }
    std::stable_sort(all.begin(), all.end(), [](const std::pair<int, Elem>& x, const std::pair<int, Elem>& y) {
      return x.first != y.first ? x.first < y.first : x.second.state < y.second.state;
    });
    size_t cur = 0;
    while (cur < all.size() && !failed) {
      const int il = all[cur].first;
      sub.clear();
      while (cur < all.size() && all[cur].first == il) {        // one Element per destination state, weights added (:1055-1080)
        const Elem& e = all[cur].second;
        if (!sub.empty() && sub.back().state == e.state) {
          if (sub.back().str != e.str) { failed = true; why = "FST was not functional -> not determinizable (transition)"; break; }
          sub.back().w = plus_w(sub.back().w, e.w, lg);
        } else {
          sub.push_back(e);
        }
        ++cur;
      }
      if (failed) break;
      // common output prefix and total weight, divided out (:1082-1118)
      std::vector<int> pre = rep.seq[(size_t)sub[0].str];
      float tot = sub[0].w;
      for (size_t i = 1; i < sub.size(); ++i) {
        const std::vector<int>& s2 = rep.seq[(size_t)sub[i].str];
        if (s2.size() < pre.size()) pre.resize(s2.size());
        for (size_t k = 0; k < pre.size(); ++k) if (s2[k] != pre[k]) { pre.resize(k); break; }
        tot = plus_w(tot, sub[i].w, lg);
      }
      const int common = rep.of(pre);
      for (Elem& e : sub) { e.w -= tot; e.str = rep.remove_prefix(e.str, pre.size()); }
      const int nx = subset_id(sub);
      out[(size_t)state].push_back(TArc{il, common, nx, tot});
    }
    if (max_states > 0 && (long long)out.size() > max_states) { failed = true; why = "more than max_states determinized states"; }
  }
  for (std::vector<Elem>* p : owned) delete p;
  if (failed) { set_error("fst_determinize_star: %s", why.c_str()); return nullptr; }
  // Output (:930-1040): strings become chains of arcs, all but the first with an epsilon on the input side
  const int base = (int)out.size();
  std::vector<int> src;
  std::vector<HArc> arcs;
  std::vector<float> fin((size_t)base, FINF);
  auto new_state = [&]() { fin.push_back(FINF); return (int)fin.size() - 1; };
  for (int s = 0; s < base; ++s)
    for (const TArc& t : out[(size_t)s]) {
      const std::vector<int>& seq = rep.seq[(size_t)t.str];
      if (t.nx < 0) {
        int cur = s;
        for (size_t i = 0; i < seq.size(); ++i) {
          const int nxt = new_state();
          src.push_back(cur); arcs.push_back(HArc{0, seq[i], i == 0 ? t.w : 0.f, nxt});
          cur = nxt;
        }
        fin[(size_t)cur] = seq.empty() ? t.w : 0.f;
      } else {
        int cur = s;
        for (size_t i = 0; i + 1 < seq.size(); ++i) {
          const int nxt = new_state();
          src.push_back(cur); arcs.push_back(HArc{i == 0 ? t.il : 0, seq[i], i == 0 ? t.w : 0.f, nxt});
          cur = nxt;
        }
        src.push_back(cur);
        arcs.push_back(HArc{seq.size() <= 1 ? t.il : 0, seq.empty() ? 0 : seq.back(), seq.size() <= 1 ? t.w : 0.f, t.nx});
      }
    }
  return from_lists((int)fin.size(), 0, src, arcs, fin);
}

// ---- minimize-encoded: the deterministic acceptor over labels (ilabel, olabel, quantised weight) minimised by partition
//      refinement (states start out split by their quantised final weight; a state's signature is the sorted list of
//      (label, class of the destination); repeated until the number of classes stops growing: the coarsest congruence) --------
extern "C" void* b2t_fst_minimize_encoded(const void* h, float delta) {
  if (!h) { set_error("fst_minimize_encoded: null"); return nullptr; }
  const HFst& f = *CFST(h);
  const int n = f.n();
  auto quant = [&](float w) { return w == FINF ? FINF : floorf(w / delta + 0.5f) * delta; };   // QuantizeMapper
  // encode: dense ids for (il, ol, quantised w)
  struct Key { int il, ol; float w; bool operator==(const Key& o) const { return il == o.il && ol == o.ol && w == o.w; } };
  struct KeyHash { size_t operator()(const Key& k) const { uint32_t wb; memcpy(&wb, &k.w, 4); return ((size_t)(unsigned)k.il * 0x9e3779b97f4a7c15ull) ^ ((size_t)(unsigned)k.ol << 21) ^ wb; } };
  std::unordered_map<Key, int, KeyHash> lab;
  std::vector<int> alab(f.arcs.size());
  std::vector<float> aw(f.arcs.size());
  for (size_t i = 0; i < f.arcs.size(); ++i) {
    aw[i] = quant(f.arcs[i].w);
    const Key k{f.arcs[i].il, f.arcs[i].ol, aw[i]};
    auto it = lab.find(k);
    if (it == lab.end()) it = lab.emplace(k, (int)lab.size()).first;
    alab[i] = it->second;
  }
  // determinism on the encoded label is what acceptor minimisation needs
  for (int s = 0; s < n; ++s) {
    std::vector<int> ls;
    for (int64_t i = f.row[s]; i < f.row[(size_t)s + 1]; ++i) ls.push_back(alab[(size_t)i]);
    std::sort(ls.begin(), ls.end());
    if (std::adjacent_find(ls.begin(), ls.end()) != ls.end()) { set_error("fst_minimize_encoded: state %d has two arcs with the same (ilabel, olabel, weight): determinize first", s); return nullptr; }
  }
  std::vector<int> cls((size_t)n), ncls((size_t)n);
  {
    std::unordered_map<uint32_t, int> by_final;
    for (int s = 0; s < n; ++s) {
      const float q = quant(f.fin[(size_t)s]);
      uint32_t b; memcpy(&b, &q, 4);
      auto it = by_final.find(b);
      if (it == by_final.end()) it = by_final.emplace(b, (int)by_final.size()).first;
      cls[(size_t)s] = it->second;
    }
  }
  int ncl = 0;
  for (int c : cls) ncl = std::max(ncl, c + 1);
  std::vector<std::pair<int, int>> sig;
  for (int iter = 0; iter < 100000; ++iter) {
    std::unordered_map<std::vector<int>, int, VecHash> seen;
    seen.reserve((size_t)ncl * 2 + 16);
    std::vector<int> key;
    for (int s = 0; s < n; ++s) {
      sig.clear();
      for (int64_t i = f.row[s]; i < f.row[(size_t)s + 1]; ++i) sig.push_back({alab[(size_t)i], cls[(size_t)f.arcs[(size_t)i].nx]});
      std::sort(sig.begin(), sig.end());
      key.clear();
      key.push_back(cls[(size_t)s]);
      for (auto& p : sig) { key.push_back(p.first); key.push_back(p.second); }
      auto it = seen.find(key);
      if (it == seen.end()) it = seen.emplace(key, (int)seen.size()).first;
      ncls[(size_t)s] = it->second;
    }
    const int m = (int)seen.size();
    cls.swap(ncls);
    if (m == ncl) break;
    ncl = m;
  }
  // quotient automaton: class of the start state first, the other classes in order of their first member
  std::vector<int> cid((size_t)ncl, -1), rep_state;
  int m = 0;
  cid[(size_t)cls[(size_t)f.start]] = m++; rep_state.push_back(f.start);
  for (int s = 0; s < n; ++s) if (cid[(size_t)cls[(size_t)s]] < 0) { cid[(size_t)cls[(size_t)s]] = m++; rep_state.push_back(s); }
  std::vector<int> src;
  std::vector<HArc> arcs;
  std::vector<float> fin((size_t)m, FINF);
  for (int c = 0; c < m; ++c) {
    const int s = rep_state[(size_t)c];
    fin[(size_t)c] = quant(f.fin[(size_t)s]);
    for (int64_t i = f.row[s]; i < f.row[(size_t)s + 1]; ++i) {
      const HArc& a = f.arcs[(size_t)i];
      src.push_back(c); arcs.push_back(HArc{a.il, a.ol, aw[(size_t)i], cid[(size_t)cls[(size_t)a.nx]]});
    }
  }
  return from_lists(m, 0, src, arcs, fin);
}

// ---- OpenFST "vector" / "standard" container (fst/fst.h FstHeader::Write, fst/vector-fst.h VectorFstImpl::Write) ---------------
extern "C" void* b2t_fst_read_openfst(const char* path) {
  FILE* fp = path ? fopen(path, "rb") : nullptr;
  if (!fp) { set_error("fst_read_openfst: cannot open %s", path ? path : "(null)"); return nullptr; }
  auto fail = [&](const char* why) { fclose(fp); set_error("fst_read_openfst: %s: %s", path, why); return (void*)nullptr; };
  int32_t magic = 0;
  if (fread(&magic, 4, 1, fp) != 1 || magic != 2125659606) return fail("not an OpenFST binary");
  auto rstr = [&](std::string& s) { int32_t n = 0; if (fread(&n, 4, 1, fp) != 1 || n < 0 || n > 4096) return false; s.resize((size_t)n); return n == 0 || fread(&s[0], 1, (size_t)n, fp) == (size_t)n; };
  std::string ftype, atype;
  if (!rstr(ftype) || !rstr(atype)) return fail("truncated header");
  if (ftype != "vector" || atype != "standard") return fail("fst type / arc type not supported (need vector / standard: fstconvert --fst_type=vector)");
  int32_t version, flags; uint64_t props; int64_t start, ns, na;
  if (fread(&version, 4, 1, fp) != 1 || fread(&flags, 4, 1, fp) != 1 || fread(&props, 8, 1, fp) != 1 || fread(&start, 8, 1, fp) != 1 ||
      fread(&ns, 8, 1, fp) != 1 || fread(&na, 8, 1, fp) != 1) return fail("truncated header");
  if (flags & 3) return fail("embedded symbol tables are not supported (the recipe compiles with --keep_isymbols=false)");
  if (ns <= 0 || ns > INT32_MAX || start < 0 || start >= ns) return fail("bad state count / start state");
  HFst* f = new HFst;
  f->start = (int)start;
  f->fin.assign((size_t)ns, FINF);
  f->row.assign((size_t)ns + 1, 0);
  if (na > 0) f->arcs.reserve((size_t)na);
  for (int64_t s = 0; s < ns; ++s) {
    float fw; int64_t n;
    if (fread(&fw, 4, 1, fp) != 1 || fread(&n, 8, 1, fp) != 1 || n < 0) { delete f; return fail("truncated state"); }
    f->fin[(size_t)s] = fw;
    const size_t at = f->arcs.size();
    f->arcs.resize(at + (size_t)n);
    static_assert(sizeof(HArc) == 16, "arc record");
    if (n > 0 && fread(&f->arcs[at], 16, (size_t)n, fp) != (size_t)n) { delete f; return fail("truncated arcs"); }
    f->row[(size_t)s + 1] = (int64_t)f->arcs.size();
  }
  fclose(fp);
  return f;
}

extern "C" int b2t_fst_write_openfst(const void* h, const char* path) {
  B2T_REQUIRE(h && path, "fst_write_openfst: null argument");
  const HFst& f = *CFST(h);
  FILE* fp = fopen(path, "wb");
  B2T_REQUIRE(fp != nullptr, "fst_write_openfst: cannot open %s", path);
  auto wstr = [&](const char* s) { const int32_t n = (int32_t)strlen(s); fwrite(&n, 4, 1, fp); fwrite(s, 1, (size_t)n, fp); };
  const int32_t magic = 2125659606, version = 2, flags = 0; const uint64_t props = 0;
  const int64_t start = f.start, ns = f.n(), na = (int64_t)f.arcs.size();
  fwrite(&magic, 4, 1, fp); wstr("vector"); wstr("standard");
  fwrite(&version, 4, 1, fp); fwrite(&flags, 4, 1, fp); fwrite(&props, 8, 1, fp); fwrite(&start, 8, 1, fp); fwrite(&ns, 8, 1, fp); fwrite(&na, 8, 1, fp);
  for (int64_t s = 0; s < ns; ++s) {
    const float fw = f.fin[(size_t)s]; const int64_t n = f.row[(size_t)s + 1] - f.row[(size_t)s];
    fwrite(&fw, 4, 1, fp); fwrite(&n, 8, 1, fp);
    if (n > 0) fwrite(&f.arcs[(size_t)f.row[(size_t)s]], 16, (size_t)n, fp);
  }
  fclose(fp);
  return 0;
}

// ---- cost of a word sequence through a grammar (BrainSpeechDecoder::LatticeRescore's composition, brain_speech_decoder.cc:44-58):
//      arcs carrying `backoff_label` on the input side may be taken freely; cheapest path + final cost.  The grammar must be
//      arc-sorted by ilabel (b2t_fst_arcsort): a (state, label) lookup is one bisection -------------------------------------------
extern "C" double b2t_fst_grammar_score(const void* h, const int32_t* words, int n_words, int backoff_label) {
  if (!h || (n_words > 0 && !words)) { set_error("fst_grammar_score: null argument"); return NAN; }
  const HFst& f = *CFST(h);
  auto range = [&](int s, int label, int64_t& lo, int64_t& hi) {
    const HArc* p0 = f.arcs.data() + f.row[s];
    const HArc* p1 = f.arcs.data() + f.row[(size_t)s + 1];
    lo = std::lower_bound(p0, p1, label, [](const HArc& x, int l) { return x.il < l; }) - f.arcs.data();
    hi = std::upper_bound(p0, p1, label, [](int l, const HArc& x) { return l < x.il; }) - f.arcs.data();
  };
  std::unordered_map<int, double> cur, nxt;
  auto close = [&](std::unordered_map<int, double>& d) {
    std::vector<int> st;
    for (auto& kv : d) st.push_back(kv.first);
    while (!st.empty()) {
      const int s = st.back(); st.pop_back();
      const double c = d[s];
      int64_t lo, hi;
      range(s, backoff_label, lo, hi);
      for (int64_t i = lo; i < hi; ++i) {
        const HArc& a = f.arcs[(size_t)i];
        auto it = d.find(a.nx);
        if (it == d.end() || c + a.w < it->second) { d[a.nx] = c + a.w; st.push_back(a.nx); }
      }
    }
  };
  cur[f.start] = 0.0;
  close(cur);
  for (int k = 0; k < n_words; ++k) {
    nxt.clear();
    for (auto& kv : cur) {
      int64_t lo, hi;
      range(kv.first, words[k], lo, hi);
      for (int64_t i = lo; i < hi; ++i) {
        const HArc& a = f.arcs[(size_t)i];
        auto it = nxt.find(a.nx);
        if (it == nxt.end() || kv.second + a.w < it->second) nxt[a.nx] = kv.second + a.w;
      }
    }
    if (nxt.empty()) return INFINITY;
    close(nxt);
    cur.swap(nxt);
  }
  double best = INFINITY;
  for (auto& kv : cur) if (f.fin[(size_t)kv.first] != FINF) best = std::min(best, kv.second + (double)f.fin[(size_t)kv.first]);
  return best;
}

// ---- BrainSpeechDecoder::Rescore (brain_speech_decoder.cc:47-101) as lattice composition ------------------------------------
// The reference composes the lattice with the grammar that is inside the decode graph at scale -1, determinises, composes with
// the rescoring grammar at scale +1, determinises, and lists the n shortest paths.  Per word sequence W that amounts to
//     graph(W) - min over routes of G_old(W) + min over routes of G_new(W),   acoustic(W) unchanged
// (DeterminizeLattice keeps, per word sequence, the cheapest (path, route) pair; a route may take back-off arcs anywhere).
// Done here on EVERY word sequence of the lattice, not on a list: each grammar is determinised lazily in the tropical semiring
// along the words the lattice actually holds (a determinised state is the weighted set of grammar states a word prefix can be
// in, closed under back-off arcs and normalised to residuals; for an n-gram grammar that is a history's back-off chain, so
// the sets stay tiny and few), the lattice is multiplied with the two deterministic machines (product state = lattice
// state x old set x new set; a lattice path then has exactly ONE route in each, so its exchanged cost is a plain sum and the
// min / max conflict of subtracting a minimum disappears), and the n-best distinct word sequences of the product come from
// the same best-first subset construction as the first pass (lattice.cpp), ranked by the new costs and confined to the
// word sequences within `beam` of the best on the OLD costs (what GetLattice left in lat_ before Rescore() ran).
namespace b2t {
namespace {

// Open-addressing table of 64-bit keys -> 32-bit payload index (the product construction does ~10^7 lookups for a wide lattice:
// std::unordered_map's node allocations and pointer chasing were most of its time).
struct FlatMap {
  std::vector<uint64_t> keys; std::vector<int> vals; size_t n = 0, mask = 0;
  static uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }
  explicit FlatMap(size_t cap = 1 << 12) { size_t c = 16; while (c < cap) c <<= 1; keys.assign(c, ~0ull); vals.assign(c, -1); mask = c - 1; }
  void grow() {
    FlatMap g(2 * (mask + 1));
    for (size_t i = 0; i <= mask; ++i) if (keys[i] != ~0ull) g.put(keys[i], vals[i]);
    keys.swap(g.keys); vals.swap(g.vals); mask = g.mask; n = g.n;
  }
  int* find(uint64_t k) {                           // key ~0 is reserved (never produced: ids and labels are < 2^31)
    for (size_t i = mix(k) & mask;; i = (i + 1) & mask) {
      if (keys[i] == k) return &vals[i];
      if (keys[i] == ~0ull) return nullptr;
    }
  }
  void put(uint64_t k, int v) {
    if (2 * (n + 1) > mask + 1) grow();
    for (size_t i = mix(k) & mask;; i = (i + 1) & mask) {
      if (keys[i] == ~0ull) { keys[i] = k; vals[i] = v; ++n; return; }
      if (keys[i] == k) { vals[i] = v; return; }
    }
  }
};

struct LmDet {
  const HFst& g;
  const int backoff;
  // determinised states: sorted (grammar state, residual) lists back to back in `pool`; residuals compared at float precision
  struct Ref { size_t off; int n; double fin; };
  std::vector<Ref> states;
  std::vector<std::pair<int, double>> pool;
  std::unordered_multimap<uint64_t, int> index;                            // content hash -> state id (verified against the pool)
  FlatMap memo;                                                            // (state id, word) -> slot in memo_val
  std::vector<std::pair<int, double>> memo_val, scratch;
  LmDet(const HFst& f, int backoff_label) : g(f), backoff(backoff_label), memo(1 << 14) {}
  size_t memo_size() const { return memo_val.size(); }

  void range(int s, int label, int64_t& lo, int64_t& hi) const {
    const HArc* p0 = g.arcs.data() + g.row[s];
    const HArc* p1 = g.arcs.data() + g.row[(size_t)s + 1];
    lo = std::lower_bound(p0, p1, label, [](const HArc& x, int l) { return x.il < l; }) - g.arcs.data();
    hi = std::upper_bound(p0, p1, label, [](int l, const HArc& x) { return l < x.il; }) - g.arcs.data();
  }
  // closure of v under back-off arcs, normalisation, interning: -> (id, the minimum that was taken out).  v is small (a
  // history's back-off chain): linear searches, no allocation besides the pool's growth.
  std::pair<int, double> intern(std::vector<std::pair<int, double>>& v) {
    auto relax = [&](int st, double c) {
      for (auto& kv : v) if (kv.first == st) { if (c < kv.second) { kv.second = c; return true; } return false; }
      v.push_back({st, c});
      return true;
    };
    {                                                                        // duplicates of the input: keep the cheapest
      size_t w = 0;
      for (size_t i = 0; i < v.size(); ++i) {
        bool dup = false;
        for (size_t k = 0; k < w; ++k) if (v[k].first == v[i].first) { v[k].second = std::min(v[k].second, v[i].second); dup = true; break; }
        if (!dup) v[w++] = v[i];
      }
      v.resize(w);
    }
    for (bool moved = true; moved;) {                                        // back-off chains are short: to the fixed point
      moved = false;
      for (size_t i = 0; i < v.size(); ++i) {
        const int s = v[i].first; const double c = v[i].second;
        int64_t lo, hi;
        range(s, backoff, lo, hi);
        for (int64_t a = lo; a < hi; ++a) moved |= relax(g.arcs[(size_t)a].nx, c + (double)g.arcs[(size_t)a].w);
      }
    }
    std::sort(v.begin(), v.end());
    double m = INFINITY, fin = INFINITY;
    for (auto& kv : v) m = std::min(m, kv.second);
    uint64_t h = 1469598103934665603ull;
    for (auto& kv : v) {
      kv.second -= m;
      const float r = (float)kv.second;
      uint32_t rb; memcpy(&rb, &r, 4);
      h = (h ^ (uint64_t)(uint32_t)kv.first) * 1099511628211ull; h = (h ^ rb) * 1099511628211ull;
      if (g.fin[(size_t)kv.first] != FINF) fin = std::min(fin, kv.second + (double)g.fin[(size_t)kv.first]);
    }
    auto rng = index.equal_range(h);
    for (auto it = rng.first; it != rng.second; ++it) {
      const Ref& r = states[(size_t)it->second];
      if (r.n != (int)v.size()) continue;
      bool same = true;
      for (int k = 0; k < r.n && same; ++k) same = pool[r.off + k].first == v[k].first && (float)pool[r.off + k].second == (float)v[k].second;
      if (same) return {it->second, m};
    }
    states.push_back(Ref{pool.size(), (int)v.size(), fin});
    pool.insert(pool.end(), v.begin(), v.end());
    index.emplace(h, (int)states.size() - 1);
    return {(int)states.size() - 1, m};
  }
  int start() { scratch.assign(1, {g.start, 0.0}); return intern(scratch).first; }
  std::pair<int, double> step(int id, int word) {
    const uint64_t k = ((uint64_t)(uint32_t)id << 32) | (uint32_t)word;
    if (int* slot = memo.find(k)) return memo_val[(size_t)*slot];
    scratch.clear();
    const Ref r0 = states[(size_t)id];
    for (int q = 0; q < r0.n; ++q) {
      const std::pair<int, double> kv = pool[r0.off + q];
      int64_t lo, hi;
      range(kv.first, word, lo, hi);
      for (int64_t i = lo; i < hi; ++i) scratch.push_back({g.arcs[(size_t)i].nx, kv.second + (double)g.arcs[(size_t)i].w});
    }
    std::pair<int, double> r{-1, INFINITY};
    if (!scratch.empty()) r = intern(scratch);
    memo.put(k, (int)memo_val.size());
    memo_val.push_back(r);
    return r;
  }
};

struct TripleHash {
  size_t operator()(const std::array<int, 3>& t) const {
    size_t h = 1469598103934665603ull;
    for (int x : t) { h ^= (size_t)(unsigned)x; h *= 1099511628211ull; }
    return h;
  }
};

}  // namespace
}  // namespace b2t

int b2t_lattice_nbest_core(int n_states, int start, int n_arcs, const int32_t* src, const int32_t* dst,
                           const int32_t* ilabel, const int32_t* olabel, const float* graph, const float* acoustic,
                           const float* delta, int n_final, const int32_t* final_state, const float* final_cost,
                           const float* final_delta, int nbest, float beam,
                           int32_t* out_words, int32_t* w_off, int w_cap, int32_t* out_ali, int32_t* a_off, int a_cap,
                           float* costs);

// ---- Rescore() on the DETERMINISED lattice (round 4) --------------------------------------------------------------------------
// The reference composes `lat_` with the grammars: the lattice GetLattice has determinised over words (one path per word
// sequence; lattice-faster-decoder.cc:193-213 -> DeterminizeLatticePruned; brain_speech_decoder.cc:47-101).  Composing the RAW
// token lattice instead multiplies every alignment variant of a word sequence by the grammar histories that reach it (3-28x the
// arcs: 5 M product arcs and 0.84 s for a 175 k-arc lattice).  So: (1) subset construction over the word labels with residual
// weights in the tropical semiring -- a determinised state is a sorted set of (lattice state, residual (total, graph, acoustic))
// (in the lattice's topological order) with the cheapest entry at (0, 0, 0), equal sets are ONE state, the normalisation offsets are the arc weights -- pruned with
// the lattice beam through the exact backward costs; states are expanded in the order of their earliest lattice state, so every
// predecessor has been expanded and a state's forward cost is final when it is popped; (2) the determinised word lattice x both
// lazily determinised grammars (LmDet): a DETERMINISTIC product, every path a distinct word sequence; (3) its paths in order of
// the NEW cost (best-first over partial paths with the exact backward cost as bound) among those within the beam on the OLD cost.
// Alignments: per entry of a determinised arc's target, the entry of its source it came from and the input labels on the way --
// computed only for the arcs an answer passes through (DetRescore::trace repeats such an arc's closure with the tracking on: round 5);
// a hypothesis' alignment is read back from its final entry.
namespace b2t {
namespace {

struct DetRescore {
  // Kept lattice states are renumbered by their topological rank (round 5): a state IS its rank, so the closures' heap holds
  // states, the per-state arrays (beta, fin, slot, arc offsets) are dense and a closure -- which moves forward a frame or two --
  // works in a narrow window of them; epsilon-output arcs and word arcs are stored apart (the closure reads only the former, the
  // gather only the latter).  The instance is kept per host thread and reused: a 175 k-arc lattice needs ~40 MB of these vectors,
  // and fresh ones are mapped and unmapped page by page on every call.
  struct RArc { int lab, nlab, ol, dst; double g, a; };      // lab / nlab: its input labels (non-epsilon) in `labels`: single-exit states are absorbed into the arcs that enter them (setup)
  struct Ent { int s; double tot, gr, ac; };
  struct Key { int s; float t, g; };                          // what makes two entries THE SAME: the lattice state, total and graph residual at float precision
  struct St { size_t off; int n; double alpha; int minrank; double fin_tot, fin_gr, fin_ac; int fin_ent; bool queued; size_t wl_off; int wl_n; };
  struct DArc { int src, dst, word; double tot, gr, ac; };
  struct ANode { int parent, label; };
  int n_kept = 0, start_r = -1;
  struct SInfo { double beta; int slot, eoff; };             // what a closure needs of a state, in one 16-byte record (slot: its index in the working subset, -1 = absent)
  std::vector<SInfo> si;                                     // the template: every thread's context works on a copy (Ctx::si)
  std::vector<unsigned char> sflag;                          // bit 0: the state has word arcs; bit 1: it is final
  std::vector<int> woff, orig, wl;                     // orig: the lattice's own id of a state (ties between equal costs are broken by it); wl: per determinised state, its entries that have word arcs
  std::vector<int> eoff;
  std::vector<RArc> earc, warc;
  std::vector<int> labels;
  std::vector<double> fin, beta;
  std::vector<St> st;
  std::vector<Ent> ent;
  std::vector<Key> ekey;                                     // ent's identity keys, same indexing (a candidate state is compared with one memcmp)
  std::vector<DArc> darc;
  std::vector<int> tr_src, tr_ali, start_ali;                // back pointers of the traced arcs (trace()), of the start state's entries
  std::vector<size_t> tr_off;
  double limit = 0;
  std::vector<ANode> ali;
  std::vector<uint64_t> ikey; std::vector<int> ival; size_t imask = 0;   // content hash -> determinised state (open addressing; equal hashes sit in successive slots)
  // set-up scratch
  struct Raw { int il, ol, dst; float gr, ac; };
  std::vector<Raw> raw;
  std::vector<int> roff, coff, po, pending, order, rank;
  std::vector<char> keep;
  std::vector<RArc> carc;
  std::vector<double> fin_raw;

  bool setup(int n, int start, int n_arcs, const int32_t* src, const int32_t* dst, const int32_t* il, const int32_t* ol, const float* gr,
             const float* ac, int n_final, const int32_t* fs, const float* fc) {
    fin_raw.assign((size_t)n, INFINITY);
    for (int i = 0; i < n_final; ++i) fin_raw[(size_t)fs[i]] = std::min(fin_raw[(size_t)fs[i]], (double)fc[i]);
    const auto T0 = std::chrono::steady_clock::now();
    // raw adjacency: the arcs as records in source order (a chain hop below then reads one cache line, not five arrays)
    roff.assign((size_t)n + 1, 0);
    for (int i = 0; i < n_arcs; ++i) ++roff[(size_t)src[i] + 1];
    for (int s = 0; s < n; ++s) roff[(size_t)s + 1] += roff[s];
    raw.resize((size_t)n_arcs);
    { po.assign(roff.begin(), roff.end() - 1); for (int i = 0; i < n_arcs; ++i) raw[(size_t)po[src[i]]++] = Raw{il[i], ol[i], dst[i], gr[i], ac[i]}; }
    const auto T1 = std::chrono::steady_clock::now();
    // Chain contraction: a state with ONE outgoing arc whose output is epsilon (a token that merely lives on through a frame),
    // neither final nor the start, is absorbed into each of its incoming arcs -- labels concatenated, costs added.  The token
    // lattice is mostly such chains (2.1 arcs per state); the epsilon closures of the subset construction then walk two thirds
    // of the states.  (Until round 5 only states with ONE incoming arc were absorbed: 67.5 k instead of 53 k kept states of 82 k.)
    auto link = [&](int v) { return v != start && roff[(size_t)v + 1] - roff[v] == 1 && fin_raw[(size_t)v] == INFINITY && raw[(size_t)roff[v]].ol == 0; };
    coff.assign((size_t)n + 1, 0);
    keep.assign((size_t)n, 0);
    for (int v = 0; v < n; ++v) { keep[(size_t)v] = !link(v); if (keep[(size_t)v]) coff[(size_t)v + 1] = roff[(size_t)v + 1] - roff[v]; }
    for (int s = 0; s < n; ++s) coff[(size_t)s + 1] += coff[s];
    carc.resize((size_t)coff[(size_t)n]);
    labels.clear();
    pending.assign((size_t)n, 0);
    {
      for (int u = 0; u < n; ++u) {
        if (!keep[(size_t)u]) continue;
        int w = coff[u];
        for (int k = roff[u]; k < roff[(size_t)u + 1]; ++k) {
          const Raw* r = &raw[(size_t)k];
          RArc x{(int)labels.size(), 0, r->ol, r->dst, (double)r->gr, (double)r->ac};
          if (r->il) labels.push_back(r->il);
          int guard = 0;
          while (!keep[(size_t)x.dst] && guard++ < n) {                       // follow the chain
            r = &raw[(size_t)roff[x.dst]];
            if (r->il) labels.push_back(r->il);
            x.g += (double)r->gr; x.a += (double)r->ac; x.dst = r->dst;
          }
          if (!keep[(size_t)x.dst]) return false;                             // a cycle of chain links
          x.nlab = (int)labels.size() - x.lab;
          carc[(size_t)w++] = x;
          ++pending[(size_t)x.dst];
        }
      }
    }
    const auto T2 = std::chrono::steady_clock::now();
    order.clear(); rank.assign((size_t)n, -1);
    n_kept = 0;
    for (int s = 0; s < n; ++s) if (keep[(size_t)s]) { ++n_kept; if (!pending[(size_t)s]) order.push_back(s); }
    for (size_t h = 0; h < order.size(); ++h) {
      const int s = order[h]; rank[(size_t)s] = (int)h;
      for (int k = coff[s]; k < coff[(size_t)s + 1]; ++k) if (--pending[(size_t)carc[(size_t)k].dst] == 0) order.push_back(carc[(size_t)k].dst);
    }
    if ((int)order.size() != n_kept) return false;                 // a cycle: the caller falls back to the raw composition
    const auto T3 = std::chrono::steady_clock::now();
    // the renumbered form: state = rank; a state's arcs keep their order within each of the two classes
    const size_t nk = (size_t)n_kept;
    eoff.assign(nk + 1, 0); woff.assign(nk + 1, 0);
    orig.assign(order.begin(), order.end());
    fin.resize(nk); beta.resize(nk);
    earc.resize(carc.size()); warc.resize(carc.size());          // (upper bounds; cut to size below)
    {
      int pe = 0, pw = 0;
      for (size_t r = 0; r < nk; ++r) {
        const int s = order[r];
        fin[r] = fin_raw[(size_t)s];
        for (int k = coff[s]; k < coff[(size_t)s + 1]; ++k) {
          RArc x = carc[(size_t)k];
          x.dst = rank[(size_t)x.dst];
          if (x.ol == 0) earc[(size_t)pe++] = x; else warc[(size_t)pw++] = x;
        }
        eoff[r + 1] = pe; woff[r + 1] = pw;
      }
      earc.resize((size_t)pe); warc.resize((size_t)pw);
    }
    const auto T4 = std::chrono::steady_clock::now();
    for (size_t h = nk; h-- > 0;) {
      double b = fin[h];
      for (int k = eoff[h]; k < eoff[h + 1]; ++k) b = std::min(b, beta[(size_t)earc[(size_t)k].dst] + (earc[(size_t)k].g + earc[(size_t)k].a));
      for (int k = woff[h]; k < woff[h + 1]; ++k) b = std::min(b, beta[(size_t)warc[(size_t)k].dst] + (warc[(size_t)k].g + warc[(size_t)k].a));
      beta[h] = b;
    }
    const auto T5 = std::chrono::steady_clock::now();
    si.resize(nk + 1);
    sflag.resize(nk);
    for (size_t r = 0; r < nk; ++r) sflag[r] = (unsigned char)((woff[r + 1] > woff[r] ? 1 : 0) | (fin[r] != INFINITY ? 2 : 0));
    for (size_t r = 0; r <= nk; ++r) si[r] = SInfo{r < nk ? beta[r] : INFINITY, -1, eoff[r]};
    start_r = rank[(size_t)start];
    st.clear(); ent.clear(); ekey.clear(); wl.clear(); darc.clear(); tr_src.clear(); tr_ali.clear(); tr_off.clear(); start_ali.clear(); ali.clear();
    if (ikey.size() < 4096) { ikey.resize(4096); ival.resize(4096); }
    std::fill(ikey.begin(), ikey.end(), 0ull); imask = ikey.size() - 1;
    n_closure_states = n_entries_expanded = 0; n_spec = n_respec = n_batches = 0;
    if (getenv("B2T_LAT_TIMING")) { auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
      fprintf(stderr, "det setup: adjacency %.2f, contraction %.2f, kahn %.2f, renumber %.2f, beta %.2f, clear %.2f ms (%d kept states, %zu + %zu arcs)\n", ms(T0, T1), ms(T1, T2), ms(T2, T3), ms(T3, T4), ms(T4, T5), ms(T5, std::chrono::steady_clock::now()), n_kept, earc.size(), warc.size()); }
    return start_r >= 0;
  }
  // An alignment node stands for ONE arc behind `parent`: `label` is the arc's index in earc, or ~index in warc; its input labels
  // (labels[lab .. lab + nlab)) are spelled out only when a hypothesis is read back (a node per label was 2-3x the nodes, written
  // on every relaxation, for the 100 paths that are ever read).
  int push_ali(int parent, int arc_ref) { ali.push_back(ANode{parent, arc_ref}); return (int)ali.size() - 1; }
  void read_ali(int n, std::vector<int>& out_rev) const {        // appends the labels from node n back to the chain's head, LAST label first
    for (; n >= 0; n = ali[(size_t)n].parent) {
      const int r = ali[(size_t)n].label;
      const RArc& x = r >= 0 ? earc[(size_t)r] : warc[(size_t)~r];
      for (int k = x.nlab - 1; k >= 0; --k) out_rev.push_back(labels[(size_t)x.lab + (size_t)k]);
    }
  }

  // What ONE thread needs to expand a determinised state: its own copy of the per-state records (the slot marks are per thread),
  // the working subset, and an arena for the expansions it produces.  ctx[0] is the calling thread's.
  struct Tr { int ol, dst; double tot, gr, ac; };
  struct Group { int word; size_t off; int n; double t, g, a; uint64_t h; };   // one word arc out of a state: its target's entries (normalised, canonical order) at r_ent[off .. off + n)
  struct Ctx {
    std::vector<SInfo> si;
    std::vector<Ent> we; std::vector<int> wsrc, wali, pop_perm, heap;
    std::vector<Tr> trans; std::vector<std::pair<uint64_t, int>> keyed;
    std::vector<Ent> r_ent; std::vector<Key> r_key; std::vector<Group> r_grp;   // r_key: parallel to r_ent
    double m_drop = INFINITY;      // the cheapest candidate the beam refused during the current expansion (cost relative to the state's alpha, + beta)
  };
  std::vector<Ctx> ctx;
  struct Spec { int who; size_t g0; int ng; double alpha, m_drop; };   // a state expanded ahead of its turn: by ctx[who], groups r_grp[g0 .. g0 + ng), valid while alpha + m_drop > limit
  std::vector<Spec> spec;

  // epsilon-output closure of the working subset (states marked in `slot`), in topological order; `base` = forward cost of the
  // subset's reference point (for the beam), costs in `we` are relative to it.  `heap` is the pending states in DESCENDING order:
  // the next state is its back, and a state reached by an epsilon arc lies a frame or two ahead of the one being expanded, i.e.
  // near the back -- an insertion moves a handful of elements where a binary heap sifted through log n levels on every pop.
  // TRACK: also where each entry came from (wsrc: the seed's source entry; wali: its alignment chain).  The determinisation runs
  // without (a third of what a relaxation writes); trace() below repeats the closure of the few arcs an answer passes through.
  template <bool TRACK> void closure(Ctx& c, double base) {
    std::vector<int>& heap = c.heap; std::vector<Ent>& we = c.we;
    heap.clear(); c.pop_perm.clear();
    for (const Ent& e : we) heap.push_back(e.s);
    std::sort(heap.begin(), heap.end(), std::greater<int>());
    SInfo* const si = c.si.data();
    double m_drop = c.m_drop;
    while (!heap.empty()) {
      const int s = heap.back(); heap.pop_back();
      const int i = si[(size_t)s].slot;
      c.pop_perm.push_back(i);                                   // topological order = the canonical order of a stored state's entries
      const Ent e = we[(size_t)i]; const int esrc = TRACK ? c.wsrc[(size_t)i] : -1, eali = TRACK ? c.wali[(size_t)i] : -1;
      for (int k = si[(size_t)s].eoff, k1 = si[(size_t)s + 1].eoff; k < k1; ++k) {
        const RArc& a = earc[(size_t)k];
        const double nt = e.tot + a.g + a.a;
        SInfo& sd = si[(size_t)a.dst];
        if (base + nt + sd.beta > limit) { if (nt + sd.beta < m_drop) m_drop = nt + sd.beta; continue; }
        const int j = sd.slot;
        if (j >= 0 && !(nt < we[(size_t)j].tot)) continue;
        const Ent ne{a.dst, nt, e.gr + a.g, e.ac + a.a};
        const int na = TRACK && a.nlab ? push_ali(eali, k) : eali;
        if (j < 0) {
          sd.slot = (int)we.size(); we.push_back(ne);
          if (TRACK) { c.wsrc.push_back(esrc); c.wali.push_back(na); }
          __builtin_prefetch(earc.data() + sd.eoff);
          size_t p = heap.size(); heap.push_back(a.dst);
          while (p > 0 && heap[p - 1] < a.dst) { heap[p] = heap[p - 1]; --p; }
          heap[p] = a.dst;
        } else { we[(size_t)j] = ne; if (TRACK) { c.wsrc[(size_t)j] = esrc; c.wali[(size_t)j] = na; } }
      }
    }
    c.m_drop = m_drop;
  }
  // The working subset of `c` (closed) -> one Group behind c.r_grp: slots released, entries normalised (the cheapest at 0; equal
  // costs: the lattice's lower state id), in canonical order, hashed.  Touches nothing shared.
  void seal(Ctx& c, int word) {
    std::vector<Ent>& we = c.we;
    for (const Ent& e : we) c.si[(size_t)e.s].slot = -1;
    size_t b = 0;
    for (size_t i = 1; i < we.size(); ++i)
      if (we[i].tot < we[b].tot || (we[i].tot == we[b].tot && orig[(size_t)we[i].s] < orig[(size_t)we[b].s])) b = i;
    const double t = we[b].tot, g = we[b].gr, a = we[b].ac;
    uint64_t h = 1469598103934665603ull;
    const size_t off = c.r_ent.size();
    for (int p : c.pop_perm) {                                   // (the closure popped every state exactly once, in topological order)
      Ent e = we[(size_t)p];
      e.tot -= t; e.gr -= g; e.ac -= a;
      const float rt = (float)e.tot + 0.0f, rg = (float)e.gr + 0.0f;      // (+ 0: a negative zero becomes the positive one -- the keys are compared as bytes)
      uint32_t b1, b2; memcpy(&b1, &rt, 4); memcpy(&b2, &rg, 4);
      h = (h ^ (uint64_t)(uint32_t)e.s) * 1099511628211ull; h = (h ^ b1) * 1099511628211ull; h = (h ^ b2) * 1099511628211ull;
      c.r_ent.push_back(e); c.r_key.push_back(Key{e.s, rt, rg});
    }
    c.r_grp.push_back(Group{word, off, (int)we.size(), t, g, a, h | 1ull});   // (hash 0 marks an empty slot of the table)
  }
  // All word arcs out of determinised state `sd` at forward cost `alpha`: one Group each behind c.r_grp, in ascending word order
  // (within a word: in the order the state's entries and the lattice list the arcs).  Reads st / ent / wl, writes only `c`.
  void expand(Ctx& c, const St& sd, double alpha) {
    c.m_drop = INFINITY;
    c.trans.clear();
    for (int wi = 0; wi < sd.wl_n; ++wi) {
      const int i = wl[sd.wl_off + (size_t)wi];
      const Ent e = ent[sd.off + (size_t)i];
      if (alpha + e.tot + beta[(size_t)e.s] > limit) { c.m_drop = std::min(c.m_drop, e.tot + beta[(size_t)e.s]); continue; }   // (a cheaper history may have made it worth keeping: harmless)
      for (int k = woff[(size_t)e.s], k1 = woff[(size_t)e.s + 1]; k < k1; ++k) {
        const RArc& x = warc[(size_t)k];
        const double nt = e.tot + x.g + x.a;
        if (alpha + nt + beta[(size_t)x.dst] > limit) { c.m_drop = std::min(c.m_drop, nt + beta[(size_t)x.dst]); continue; }
        c.trans.push_back(Tr{x.ol, x.dst, nt, e.gr + x.g, e.ac + x.a});
      }
    }
    std::vector<std::pair<uint64_t, int>>& keyed = c.keyed;
    keyed.resize(c.trans.size());
    for (size_t q = 0; q < c.trans.size(); ++q) keyed[q] = {((uint64_t)(uint32_t)c.trans[q].ol << 32) | (uint64_t)q, (int)q};
    std::sort(keyed.begin(), keyed.end());
    for (size_t g0 = 0; g0 < keyed.size();) {
      size_t g1 = g0;
      const int ol = c.trans[(size_t)keyed[g0].second].ol;
      while (g1 < keyed.size() && c.trans[(size_t)keyed[g1].second].ol == ol) ++g1;
      c.we.clear();
      for (size_t q = g0; q < g1; ++q) {
        const Tr& tr = c.trans[(size_t)keyed[q].second];
        const int j = c.si[(size_t)tr.dst].slot;
        if (j >= 0 && !(tr.tot < c.we[(size_t)j].tot)) continue;
        const Ent ne{tr.dst, tr.tot, tr.gr, tr.ac};
        if (j < 0) { c.si[(size_t)tr.dst].slot = (int)c.we.size(); c.we.push_back(ne); }
        else c.we[(size_t)j] = ne;
      }
      closure<false>(c, alpha);
      seal(c, ol);
      g0 = g1;
    }
  }
  // Back pointers of determinised arc `id` (per entry of its target, in the stored order: the source state's entry it came from
  // and its alignment chain), computed on demand: the arc's seeds are gathered and closed again exactly as expand() did it -- same
  // entries in the same order, the source's forward cost is final since it was expanded -- this time with the tracking on.
  size_t trace(int id) {
    if (tr_off.size() != darc.size()) tr_off.assign(darc.size(), (size_t)-1);
    if (tr_off[(size_t)id] != (size_t)-1) return tr_off[(size_t)id];
    Ctx& c = ctx[0];
    const DArc& da = darc[(size_t)id];
    const St sd = st[(size_t)da.src];
    c.we.clear(); c.wsrc.clear(); c.wali.clear();
    for (int wi = 0; wi < sd.wl_n; ++wi) {
      const int i = wl[sd.wl_off + (size_t)wi];
      const Ent e = ent[sd.off + (size_t)i];
      if (sd.alpha + e.tot + beta[(size_t)e.s] > limit) continue;
      for (int k = woff[(size_t)e.s], k1 = woff[(size_t)e.s + 1]; k < k1; ++k) {
        const RArc& x = warc[(size_t)k];
        if (x.ol != da.word) continue;
        const double nt = e.tot + x.g + x.a;
        if (sd.alpha + nt + beta[(size_t)x.dst] > limit) continue;
        const int j = c.si[(size_t)x.dst].slot;
        if (j >= 0 && !(nt < c.we[(size_t)j].tot)) continue;
        const Ent ne{x.dst, nt, e.gr + x.g, e.ac + x.a};
        const int na = x.nlab ? push_ali(-1, ~k) : -1;
        if (j < 0) { c.si[(size_t)x.dst].slot = (int)c.we.size(); c.we.push_back(ne); c.wsrc.push_back(i); c.wali.push_back(na); }
        else { c.we[(size_t)j] = ne; c.wsrc[(size_t)j] = i; c.wali[(size_t)j] = na; }
      }
    }
    closure<true>(c, sd.alpha);
    for (const Ent& e : c.we) c.si[(size_t)e.s].slot = -1;
    const size_t off = tr_src.size();
    for (int p : c.pop_perm) { tr_src.push_back(c.wsrc[(size_t)p]); tr_ali.push_back(c.wali[(size_t)p]); }
    return tr_off[(size_t)id] = off;
  }
  void index_grow() {
    std::vector<uint64_t> k2(ikey.size() * 2, 0ull); std::vector<int> v2(ikey.size() * 2, -1);
    const size_t m2 = k2.size() - 1;
    for (size_t i = 0; i < ikey.size(); ++i)
      if (ikey[i]) { size_t p = (size_t)(ikey[i] ^ (ikey[i] >> 29)) & m2; while (k2[p]) p = (p + 1) & m2; k2[p] = ikey[i]; v2[p] = ival[i]; }
    ikey.swap(k2); ival.swap(v2); imask = m2;
  }
  // intern a sealed group (entries L[0 .. gr.n)): the determinised state with exactly these entries (costs at float precision),
  // created if new; its forward cost lowered to alpha_via + gr.t if that is cheaper.  Calling thread only.
  int commit(const Group& gr, const Ent* L, const Key* K, double alpha_via) {
    const uint64_t h = gr.h;
    size_t p = (size_t)(h ^ (h >> 29)) & imask;
    for (; ikey[p]; p = (p + 1) & imask) {
      if (ikey[p] != h) continue;
      St& r = st[(size_t)ival[p]];
      if (r.n != gr.n) continue;
      if (memcmp(&ekey[r.off], K, (size_t)gr.n * sizeof(Key)) == 0) { r.alpha = std::min(r.alpha, alpha_via + gr.t); return ival[p]; }
    }
    St ns{ent.size(), gr.n, alpha_via + gr.t, L[0].s, INFINITY, 0.0, 0.0, -1, false, wl.size(), 0};   // minrank: the first state popped
    ent.insert(ent.end(), L, L + gr.n); ekey.insert(ekey.end(), K, K + gr.n);
    for (int k = 0; k < gr.n; ++k) {
      const unsigned char f = sflag[(size_t)L[k].s];           // (one byte per lattice state: this loop is the calling thread's, i.e. serial)
      if (!f) continue;
      if (f & 1) { wl.push_back(k); ++ns.wl_n; }
      if (f & 2) {
        const Ent& e = L[k];
        const double c = e.tot + fin[(size_t)e.s];
        if (c < ns.fin_tot) { ns.fin_tot = c; ns.fin_gr = e.gr + fin[(size_t)e.s]; ns.fin_ac = e.ac; ns.fin_ent = k; }
      }
    }
    st.push_back(ns);
    spec.push_back(Spec{-1, 0, 0, 0.0, 0.0});
    ikey[p] = h; ival[p] = (int)st.size() - 1;
    if (2 * st.size() > imask) index_grow();
    return (int)st.size() - 1;
  }

  // the whole determinisation; returns false if the lattice has no path.
  // n_threads > 1: states are expanded AHEAD of their turn, a batch of the queue's earliest at a time, by n_threads threads (the
  // calling one among them) that only read the shared arrays; the calling thread then takes the states in their proper order and
  // interns what was prepared.  An expansion depends on its state's forward cost alpha through the beam only: a cost lowered after
  // the expansion (a cheaper history found by a batch peer: a fifth of the states, by 0.09 on average) leaves it valid as long as
  // no candidate the beam refused would now pass (alpha + m_drop > limit); otherwise the state is expanded again in its turn.  So
  // the result is the serial one bit for bit.  (The queue holds ~1500 states when a 175 k-arc lattice is half done.)
  bool run(double beam, int n_threads) {
    failed = false; h_failed.store(false, std::memory_order_relaxed);
    HelperGuard guard{this};
    const int start = start_r;
    if (beta[(size_t)start] == INFINITY) return false;
    limit = beta[(size_t)start] + beam + 1e-4;
    if (n_threads < 1) n_threads = 1;
    if ((int)ctx.size() < n_threads) ctx.resize((size_t)n_threads);
    for (int t = 0; t < n_threads; ++t) { ctx[(size_t)t].si = si; ctx[(size_t)t].r_ent.clear(); ctx[(size_t)t].r_key.clear(); ctx[(size_t)t].r_grp.clear(); }
    spec.clear();
    Ctx& c0 = ctx[0];
    c0.we.assign(1, Ent{start, 0.0, 0.0, 0.0}); c0.wsrc.assign(1, -1); c0.wali.assign(1, -1);
    c0.si[(size_t)start].slot = 0;
    closure<true>(c0, 0.0);
    start_ali.clear();
    for (int p : c0.pop_perm) start_ali.push_back(c0.wali[(size_t)p]);
    seal(c0, 0);
    {
      const Group gr = c0.r_grp.back();
      commit(gr, c0.r_ent.data() + gr.off, c0.r_key.data() + gr.off, 0.0);              // state 0; its offset (the start's closure may hold a cheaper entry than the start itself: costs can be negative) is start_off / start_g / start_a
      start_off = gr.t; start_g = gr.g; start_a = gr.a;
      c0.r_ent.clear(); c0.r_key.clear(); c0.r_grp.clear();
    }
    typedef std::pair<int, int> QI;                        // (earliest lattice state, determinised state); min-heap in a vector
    std::vector<QI> pq, batch;
    auto qpush = [&](QI x) { pq.push_back(x); std::push_heap(pq.begin(), pq.end(), std::greater<QI>()); };
    qpush({st[0].minrank, 0}); st[0].queued = true;
    size_t outstanding = 0;                                // states expanded ahead that have not had their turn yet
    const char* mb_s = getenv("B2T_RESCORE_MIN_BATCH");           // (tests: batches on small lattices)
    const size_t BATCH = 768, MIN_BATCH = mb_s && atoi(mb_s) > 0 ? (size_t)atoi(mb_s) : 192;
    size_t unprepared = 1;                                 // queued states with nothing prepared
    while (!pq.empty()) {
      if (n_threads > 1 && unprepared >= MIN_BATCH && spec[(size_t)pq.front().second].who < 0) {
        // the next state has nothing prepared: the queue's earliest unprepared states are expanded in parallel (nothing shared
        // is written: st / ent / wl / the table rest).  With no prepared state left, the arenas start over.
        if (outstanding == 0) for (int t = 0; t < n_threads; ++t) { ctx[(size_t)t].r_ent.clear(); ctx[(size_t)t].r_key.clear(); ctx[(size_t)t].r_grp.clear(); }
        batch.clear();
        for (const QI& x : pq) if (spec[(size_t)x.second].who < 0) batch.push_back(x);
        const size_t nb = std::min(BATCH, batch.size());
        std::partial_sort(batch.begin(), batch.begin() + (long)nb, batch.end());
        batch.resize(nb);
        // publish the batch to the helpers (started with the first batch, parked on the condition variable between batches).  The
        // batch is complete when every state of it is done -- NOT when every helper has come by: one that the scheduler has not
        // run yet (the pool's other lattices keep the cores busy) takes no chunk and is not waited for.
        {
          std::unique_lock<std::mutex> lk(hm);
          dcv.wait(lk, [&] { return h_active.load(std::memory_order_acquire) == 0; });   // a late-comer of the previous batch is still looking at it (the wait releases hm)
          h_batch = batch.data(); h_nb = nb;
          h_next.store(0, std::memory_order_relaxed); h_done.store(0, std::memory_order_relaxed);
          ++h_seq;
        }
        if (helpers.empty() && !h_refused)
          for (int t = 1; t < n_threads; ++t) {
            try { helpers.emplace_back([this, t] { helper_loop(t); }); }
            catch (const std::system_error&) { h_refused = true; break; }      // no more threads to be had: the calling thread does what the missing ones would (work_chunks takes every chunk nobody else takes)
          }
        hcv.notify_all();
        work_chunks(0);
        { std::unique_lock<std::mutex> lk(hm); dcv.wait(lk, [&] { return h_done.load(std::memory_order_acquire) >= nb; }); }
        if (h_failed.load(std::memory_order_acquire)) { stop_helpers(); failed = true; return false; }   // an expand() threw (out of memory): the caller falls back
        outstanding += nb; unprepared -= nb; ++n_batches;
        n_spec += nb;
      }
      std::pop_heap(pq.begin(), pq.end(), std::greater<QI>());
      const int D = pq.back().second; pq.pop_back();
      const St sd = st[(size_t)D];
      n_entries_expanded += (size_t)sd.n;
      const Spec sp = spec[(size_t)D];
      int who = sp.who; size_t g0 = sp.g0; int ng = sp.ng;
      const size_t keep_grp = c0.r_grp.size(), keep_ent = c0.r_ent.size();
      if (who < 0) --unprepared;
      if (who >= 0) {
        --outstanding;
        if (!(sd.alpha == sp.alpha || sd.alpha + sp.m_drop > limit)) { who = -1; ++n_respec; }
      }
      if (who < 0) {                                         // its turn has come and nothing (valid) is prepared: expand it now
        who = 0; g0 = keep_grp;
        expand(c0, sd, sd.alpha);
        ng = (int)(c0.r_grp.size() - g0);
      }
      const Ctx& c = ctx[(size_t)who];
      for (int q = 0; q < ng; ++q) {
        const Group& gr = c.r_grp[g0 + (size_t)q];
        n_closure_states += (size_t)gr.n;
        const int T = commit(gr, c.r_ent.data() + gr.off, c.r_key.data() + gr.off, sd.alpha);
        darc.push_back(DArc{D, T, gr.word, gr.t, gr.g, gr.a});
        if (!st[(size_t)T].queued) { st[(size_t)T].queued = true; qpush({st[(size_t)T].minrank, T}); ++unprepared; }
      }
      if (who == 0 && g0 == keep_grp) { c0.r_grp.resize(keep_grp); c0.r_ent.resize(keep_ent); c0.r_key.resize(keep_ent); }   // groups made in turn are dropped again
    }
    stop_helpers();
    return true;
  }
  void stop_helpers() {
    if (!helpers.empty()) {
      { std::unique_lock<std::mutex> lk(hm); h_stop = true; }
      hcv.notify_all();
      for (std::thread& h : helpers) h.join();
      helpers.clear(); h_stop = false;
    }
    h_refused = false;
  }
  struct HelperGuard { DetRescore* d; ~HelperGuard() { d->stop_helpers(); } };      // (run(): an exception on the calling thread must not leave joinable threads behind)
  bool failed = false;
  void release_big() {                                           // (after an outsized lattice: the arrays go back to the allocator)
    std::vector<Ent>().swap(ent); std::vector<Key>().swap(ekey); std::vector<ANode>().swap(ali); std::vector<Ctx>().swap(ctx); std::vector<Raw>().swap(raw);
    std::vector<RArc>().swap(carc); std::vector<RArc>().swap(earc); std::vector<RArc>().swap(warc); std::vector<int>().swap(tr_src); std::vector<int>().swap(tr_ali);
  }
  // helper threads of one run(): chunks of four states of the published batch, results into the thread's own context
  std::mutex hm; std::condition_variable hcv, dcv;      // hcv: a batch is published / stop; dcv: a batch is complete / the helpers have left it
  std::vector<std::thread> helpers;
  std::atomic<size_t> h_next{0}, h_done{0};
  std::atomic<int> h_active{0};
  std::atomic<bool> h_failed{false};
  const std::pair<int, int>* h_batch = nullptr; size_t h_nb = 0; unsigned long long h_seq = 0; bool h_stop = false, h_refused = false;
  void work_chunks(int t) {
    Ctx& c = ctx[(size_t)t];
    const size_t nb = h_nb;
    for (size_t i = h_next.fetch_add(4); i < nb; i = h_next.fetch_add(4)) {
      const size_t i1 = std::min(nb, i + 4);
      try {
        for (size_t q = i; q < i1; ++q) {
          const int E = h_batch[q].second;
          const St sd = st[(size_t)E];
          const size_t g0 = c.r_grp.size();
          expand(c, sd, sd.alpha);
          spec[(size_t)E] = Spec{t, g0, (int)(c.r_grp.size() - g0), sd.alpha, c.m_drop};
        }
      } catch (...) { h_failed.store(true, std::memory_order_release); }     // (counted as done all the same: the batch must complete for run() to see the flag)
      if (h_done.fetch_add(i1 - i, std::memory_order_acq_rel) + (i1 - i) >= nb) { std::lock_guard<std::mutex> lk(hm); dcv.notify_all(); }
    }
  }
  void helper_loop(int t) {
    unsigned long long seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(hm);
        hcv.wait(lk, [&] { return h_stop || h_seq != seen; });
        if (h_stop) return;
        seen = h_seq;
        h_active.fetch_add(1, std::memory_order_acq_rel);
      }
      work_chunks(t);
      if (h_active.fetch_sub(1, std::memory_order_acq_rel) == 1) { std::lock_guard<std::mutex> lk(hm); dcv.notify_all(); }
    }
  }
  size_t n_spec = 0, n_respec = 0, n_batches = 0;
  double start_off = 0, start_g = 0, start_a = 0;
  size_t n_closure_states = 0, n_entries_expanded = 0;
};

}  // namespace
}  // namespace b2t

static int rescore_on_determinised(int n_states, int start, int n_arcs, const int32_t* src, const int32_t* dst, const int32_t* ilabel,
                                   const int32_t* olabel, const float* graph, const float* acoustic, int n_final, const int32_t* final_state,
                                   const float* final_cost, const void* g_old, const void* g_new, int backoff_label, int nbest, float beam,
                                   int32_t* out_words, int32_t* w_off, int w_cap, int32_t* out_ali, int32_t* a_off, int a_cap, float* costs,
                                   long long* stats4, bool* fell_back) {
  using namespace b2t;
  const auto t_in = std::chrono::steady_clock::now();
  *fell_back = false;
  for (int i = 0; i < n_arcs; ++i)
    if (src[i] < 0 || src[i] >= n_states || dst[i] < 0 || dst[i] >= n_states) { set_error("lattice_rescore: arc %d out of range", i); return -1; }
  for (int i = 0; i < n_final; ++i)
    if (final_state[i] < 0 || final_state[i] >= n_states) { set_error("lattice_rescore: final state %d out of range", i); return -1; }
  static thread_local DetRescore dr;                      // (vectors reused from call to call: see the struct's comment)
  struct Trim { DetRescore& d; ~Trim() { if (d.ent.capacity() * sizeof(DetRescore::Ent) + d.ali.capacity() * sizeof(DetRescore::ANode) > ((size_t)256 << 20)) d.release_big(); } } trim{dr};   // an outsized lattice does not pin its arrays to the thread
  if (!dr.setup(n_states, start, n_arcs, src, dst, ilabel, olabel, graph, acoustic, n_final, final_state, final_cost)) { *fell_back = true; return 0; }
  w_off[0] = 0; a_off[0] = 0;
  // Helper threads for the large lattices only (the batch of a call lasts as long as its largest lattice; the small ones are done
  // long before and leave their cores): B2T_RESCORE_THREADS forces a count for EVERY lattice (tests; 1 = the serial determinisation).
  const char* env_thr_s = getenv("B2T_RESCORE_THREADS");           // (read per call: tests switch it)
  const int env_threads = env_thr_s ? atoi(env_thr_s) : 0;
  const char* env_big_s = getenv("B2T_RESCORE_BIG_THREADS");      // threads of a lattice of >= 60 k arcs (default 4; 1 = serial)
  const int big_threads = env_big_s && atoi(env_big_s) > 0 ? std::min(atoi(env_big_s), 16) : 4;
  const int det_threads = env_threads > 0 ? std::min(env_threads, 16) : (n_arcs >= 60000 && std::thread::hardware_concurrency() >= 8 ? big_threads : 1);
  if (!dr.run((double)beam, det_threads)) { if (dr.failed) *fell_back = true; if (stats4) stats4[0] = stats4[1] = stats4[2] = stats4[3] = 0; return 0; }
  const auto t_det = std::chrono::steady_clock::now();
  LmDet Lo(*CFST(g_old), backoff_label), Ln(*CFST(g_new), backoff_label);
  // determinised-lattice adjacency
  const int ND = (int)dr.st.size();
  std::vector<int> doff((size_t)ND + 1, 0);
  for (const auto& a : dr.darc) ++doff[(size_t)a.src + 1];
  for (int s = 0; s < ND; ++s) doff[(size_t)s + 1] += doff[s];
  std::vector<int> darc_of(dr.darc.size());
  { std::vector<int> po(doff.begin(), doff.end() - 1); for (size_t i = 0; i < dr.darc.size(); ++i) darc_of[(size_t)po[dr.darc[i].src]++] = (int)i; }
  // product with the two grammars (deterministic)
  struct PS { int d, o, n; };
  struct PA { int src, dst, darc; double w_new, dl; };
  FlatMap pair_id(1 << 10), pid((size_t)4 * ND + 16);
  std::vector<std::array<int, 2>> pairs;
  std::vector<PS> ps;
  std::vector<PA> pa;
  auto state_of = [&](int d, int o, int n) {
    const uint64_t pk = ((uint64_t)(uint32_t)o << 32) | (uint32_t)n;
    int p;
    if (int* f = pair_id.find(pk)) p = *f; else { p = (int)pairs.size(); pairs.push_back({o, n}); pair_id.put(pk, p); }
    const uint64_t k = ((uint64_t)(uint32_t)d << 32) | (uint32_t)p;
    if (int* f = pid.find(k)) return *f;
    ps.push_back(PS{d, o, n});
    pid.put(k, (int)ps.size() - 1);
    return (int)ps.size() - 1;
  };
  state_of(0, Lo.start(), Ln.start());
  for (size_t q = 0; q < ps.size(); ++q) {
    const PS k = ps[q];
    for (int e = doff[(size_t)k.d]; e < doff[(size_t)k.d + 1]; ++e) {
      const auto& a = dr.darc[(size_t)darc_of[(size_t)e]];
      const std::pair<int, double> so = Lo.step(k.o, a.word), sn = Ln.step(k.n, a.word);
      if (so.first < 0 || sn.first < 0) continue;
      const double d = sn.second - so.second;
      const int t = state_of(a.dst, so.first, sn.first);
      pa.push_back(PA{(int)q, t, darc_of[(size_t)e], a.tot + d, d});
    }
  }
  const int NP = (int)ps.size();
  std::vector<int> poff((size_t)NP + 1, 0);
  for (const PA& a : pa) ++poff[(size_t)a.src + 1];
  for (int s = 0; s < NP; ++s) poff[(size_t)s + 1] += poff[s];          // (arcs were appended in source order: already grouped)
  std::vector<double> pfin_new((size_t)NP, INFINITY), pfin_dl((size_t)NP, 0.0);
  for (int q = 0; q < NP; ++q) {
    const auto& sd = dr.st[(size_t)ps[(size_t)q].d];
    if (sd.fin_ent < 0) continue;
    const double fo = Lo.states[(size_t)ps[(size_t)q].o].fin, fn = Ln.states[(size_t)ps[(size_t)q].n].fin;
    if (fo == INFINITY || fn == INFINITY) continue;
    pfin_new[(size_t)q] = sd.fin_tot + (fn - fo); pfin_dl[(size_t)q] = fn - fo;
  }
  // backward costs (new and old) over the product: a determinised arc strictly advances the earliest lattice state of its
  // determinised state, so the product states sorted by that rank, descending, are in reverse topological order
  std::vector<int> topo((size_t)NP);
  for (int q = 0; q < NP; ++q) topo[(size_t)q] = q;
  std::sort(topo.begin(), topo.end(), [&](int x, int y) {
    const int rx = dr.st[(size_t)ps[(size_t)x].d].minrank, ry = dr.st[(size_t)ps[(size_t)y].d].minrank;
    return rx != ry ? rx > ry : x > y;
  });
  std::vector<double> hn((size_t)NP, INFINITY), ho((size_t)NP, INFINITY);
  for (int q : topo) {
    double bn = pfin_new[(size_t)q], bo = pfin_new[(size_t)q] == INFINITY ? INFINITY : pfin_new[(size_t)q] - pfin_dl[(size_t)q];
    for (int e = poff[(size_t)q]; e < poff[(size_t)q + 1]; ++e) {
      const PA& a = pa[(size_t)e];
      bn = std::min(bn, a.w_new + hn[(size_t)a.dst]);
      bo = std::min(bo, a.w_new - a.dl + ho[(size_t)a.dst]);
    }
    hn[(size_t)q] = bn; ho[(size_t)q] = bo;
  }
  if (stats4) { stats4[0] = NP; stats4[1] = (long long)pa.size(); stats4[2] = (long long)Lo.states.size(); stats4[3] = (long long)Ln.states.size(); }
  static const bool timing = getenv("B2T_LAT_TIMING") != nullptr;
  const auto t_prod = std::chrono::steady_clock::now();
  int n_out = 0;
  if (hn[0] != INFINITY && ho[0] != INFINITY) {
    // paths in order of the new cost: best-first over partial paths, bound = cost so far + exact backward cost
    const double limit_o = dr.start_off + ho[0] + (double)beam + 1e-4;
    struct PN { int parent, parc; };
    std::vector<PN> path;
    struct It { double f, gnew, dl; int state, node; long long tie; bool done;
                bool operator<(const It& o) const { if (f != o.f) return f > o.f; if (done != o.done) return !done; return tie > o.tie; } };
    std::priority_queue<It> pq;
    long long tie = 0;
    pq.push(It{dr.start_off + hn[0], dr.start_off, 0.0, 0, -1, tie++, false});
    std::vector<int> wv, av, arcs_rev;
    while (!pq.empty() && n_out < nbest) {
      const It it = pq.top(); pq.pop();
      if (it.done) {
        // read the hypothesis back: words along the path, the alignment from the final entry through the arcs' back pointers
        arcs_rev.clear();
        for (int n = it.node; n >= 0; n = path[(size_t)n].parent) arcs_rev.push_back(path[(size_t)n].parc);
        wv.clear(); av.clear();
        double gr = 0.0, ac = 0.0;
        const auto& sdf = dr.st[(size_t)ps[(size_t)it.state].d];
        int e = sdf.fin_ent;
        gr += sdf.fin_gr + pfin_dl[(size_t)it.state]; ac += sdf.fin_ac;
        for (int parc : arcs_rev) {                                  // last arc first
          const PA& x = pa[(size_t)parc];
          const auto& da = dr.darc[(size_t)x.darc];
          wv.push_back(da.word);
          gr += da.gr + x.dl; ac += da.ac;
          const size_t bp = dr.trace(x.darc);
          dr.read_ali(dr.tr_ali[bp + (size_t)e], av);
          e = dr.tr_src[bp + (size_t)e];
        }
        dr.read_ali(dr.start_ali[(size_t)e], av);
        gr += dr.start_g; ac += dr.start_a;
        if (w_off[n_out] + (int)wv.size() > w_cap || a_off[n_out] + (int)av.size() > a_cap) { set_error("lattice_rescore: output buffers too small"); return -2; }
        std::reverse(wv.begin(), wv.end()); std::reverse(av.begin(), av.end());
        if (out_words) std::copy(wv.begin(), wv.end(), out_words + w_off[n_out]);
        if (out_ali) std::copy(av.begin(), av.end(), out_ali + a_off[n_out]);
        w_off[n_out + 1] = w_off[n_out] + (int)wv.size();
        a_off[n_out + 1] = a_off[n_out] + (int)av.size();
        costs[2 * n_out] = (float)gr; costs[2 * n_out + 1] = (float)ac;
        ++n_out;
        continue;
      }
      const int q = it.state;
      if (pfin_new[(size_t)q] != INFINITY) {
        const double old_tot = (it.gnew - it.dl) + (pfin_new[(size_t)q] - pfin_dl[(size_t)q]);
        if (old_tot <= limit_o) pq.push(It{it.gnew + pfin_new[(size_t)q], it.gnew + pfin_new[(size_t)q], it.dl + pfin_dl[(size_t)q], q, it.node, tie++, true});
      }
      for (int e = poff[(size_t)q]; e < poff[(size_t)q + 1]; ++e) {
        const PA& a = pa[(size_t)e];
        if (hn[(size_t)a.dst] == INFINITY) continue;
        const double gn = it.gnew + a.w_new, dl = it.dl + a.dl;
        if ((gn - dl) + ho[(size_t)a.dst] > limit_o) continue;          // no completion of this prefix is within the beam on the old costs
        path.push_back(PN{it.node, e});
        pq.push(It{gn + hn[(size_t)a.dst], gn, dl, a.dst, (int)path.size() - 1, tie++, false});
      }
    }
  }
  if (timing) {
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    fprintf(stderr, "lattice_rescore (determinised): %d lattice arcs -> %zu determinised states / %zu arcs (%.1f ms) -> product %d states / %zu arcs (%.1f ms), "
                    "%d hypotheses (%.1f ms); det: %d thread(s), %zu states expanded ahead of their turn in %zu batches, %zu of them again in turn; %zu closure states, %zu entries expanded\n", n_arcs, dr.st.size(), dr.darc.size(), ms(t_in, t_det), NP, pa.size(), ms(t_det, t_prod), n_out,
            ms(t_prod, std::chrono::steady_clock::now()), det_threads, dr.n_spec, dr.n_batches, dr.n_respec, dr.n_closure_states, dr.n_entries_expanded);
  }
  return n_out;
}


extern "C" int b2t_lattice_rescore_nbest_host(int n_states, int start, int n_arcs, const int32_t* src, const int32_t* dst,
                                              const int32_t* ilabel, const int32_t* olabel, const float* graph, const float* acoustic,
                                              int n_final, const int32_t* final_state, const float* final_cost,
                                              const void* g_old, const void* g_new, int backoff_label, int nbest, float beam,
                                              int32_t* out_words, int32_t* w_off, int w_cap, int32_t* out_ali, int32_t* a_off, int a_cap,
                                              float* costs, long long* stats4) {
  using namespace b2t;
  if (!g_old || !g_new || n_states <= 0 || start < 0 || start >= n_states || n_arcs < 0 || nbest <= 0) {
    set_error("lattice_rescore: bad arguments");
    return -1;
  }
  // Default: on the determinised lattice, as the reference composes lat_ (above).  B2T_RESCORE_RAW=1, and lattices with a cycle
  // of epsilon arcs, take the composition of the raw token lattice below (the first version: 3-28x the product arcs).
  {
    const char* raw_s = getenv("B2T_RESCORE_RAW");
    if (!(raw_s && atoi(raw_s) != 0)) {
      bool fell_back = false;
      const int r = rescore_on_determinised(n_states, start, n_arcs, src, dst, ilabel, olabel, graph, acoustic, n_final, final_state, final_cost,
                                            g_old, g_new, backoff_label, nbest, beam, out_words, w_off, w_cap, out_ali, a_off, a_cap, costs, stats4,
                                            &fell_back);
      if (!fell_back) return r;
    }
  }
  const auto t_in = std::chrono::steady_clock::now();
  LmDet Lo(*CFST(g_old), backoff_label), Ln(*CFST(g_new), backoff_label);
  // lattice adjacency
  std::vector<int> off((size_t)n_states + 1, 0);
  for (int i = 0; i < n_arcs; ++i) {
    if (src[i] < 0 || src[i] >= n_states || dst[i] < 0 || dst[i] >= n_states) { set_error("lattice_rescore: arc %d out of range", i); return -1; }
    ++off[(size_t)src[i] + 1];
  }
  for (int s = 0; s < n_states; ++s) off[(size_t)s + 1] += off[s];
  std::vector<int> arc_of((size_t)n_arcs);
  {
    std::vector<int> po(off.begin(), off.end() - 1);
    for (int i = 0; i < n_arcs; ++i) arc_of[(size_t)po[src[i]]++] = i;
  }
  std::vector<float> fin((size_t)n_states, FINF);
  for (int i = 0; i < n_final; ++i) fin[(size_t)final_state[i]] = std::min(fin[(size_t)final_state[i]], final_cost[i]);
  // product states in discovery order
  // product states in discovery order; (old set, new set) pairs are interned first so that a product state is ONE 64-bit key
  FlatMap pair_id(1 << 12), pid((size_t)4 * n_states);
  std::vector<std::array<int, 2>> pairs;
  std::vector<std::array<int, 3>> pst;
  auto state_of = [&](int l, int o, int n) {
    const uint64_t pk = ((uint64_t)(uint32_t)o << 32) | (uint32_t)n;
    int p;
    if (int* f = pair_id.find(pk)) p = *f; else { p = (int)pairs.size(); pairs.push_back({o, n}); pair_id.put(pk, p); }
    const uint64_t k = ((uint64_t)(uint32_t)l << 32) | (uint32_t)p;
    if (int* f = pid.find(k)) return *f;
    pst.push_back({l, o, n});
    pid.put(k, (int)pst.size() - 1);
    return (int)pst.size() - 1;
  };
  std::vector<int32_t> psrc, pdst, pil, pol;
  std::vector<float> pg, pa, pd;
  { const size_t r = (size_t)4 * n_arcs; psrc.reserve(r); pdst.reserve(r); pil.reserve(r); pol.reserve(r); pg.reserve(r); pa.reserve(r); pd.reserve(r); pst.reserve((size_t)4 * n_states); }
  state_of(start, Lo.start(), Ln.start());
  for (size_t q = 0; q < pst.size(); ++q) {
    const std::array<int, 3> k = pst[q];
    for (int e = off[(size_t)k[0]]; e < off[(size_t)k[0] + 1]; ++e) {
      const int i = arc_of[(size_t)e];
      int o = k[1], n = k[2];
      double d = 0.0;
      if (olabel[i] != 0) {
        const std::pair<int, double> so = Lo.step(k[1], olabel[i]);
        const std::pair<int, double> sn = Ln.step(k[2], olabel[i]);
        if (so.first < 0 || sn.first < 0) continue;            // a word one of the grammars does not accept: the composition drops the path
        o = so.first; n = sn.first; d = sn.second - so.second;
      }
      const int t = state_of(dst[i], o, n);
      psrc.push_back((int32_t)q); pdst.push_back(t); pil.push_back(ilabel[i]); pol.push_back(olabel[i]);
      pg.push_back((float)((double)graph[i] + d)); pa.push_back(acoustic[i]); pd.push_back((float)d);
    }
  }
  std::vector<int32_t> fst_;
  std::vector<float> fc, fd;
  for (size_t q = 0; q < pst.size(); ++q) {
    const std::array<int, 3> k = pst[q];
    if (fin[(size_t)k[0]] == FINF) continue;
    const double fo = Lo.states[(size_t)k[1]].fin, fn = Ln.states[(size_t)k[2]].fin;
    if (fo == INFINITY || fn == INFINITY) continue;
    fst_.push_back((int32_t)q); fc.push_back((float)((double)fin[(size_t)k[0]] + fn - fo)); fd.push_back((float)(fn - fo));
  }
  if (stats4) { stats4[0] = (long long)pst.size(); stats4[1] = (long long)psrc.size(); stats4[2] = (long long)Lo.states.size(); stats4[3] = (long long)Ln.states.size(); }
  w_off[0] = 0; a_off[0] = 0;
  if (fst_.empty()) return 0;
  static const bool timing = getenv("B2T_LAT_TIMING") != nullptr;
  if (timing) fprintf(stderr, "lattice_rescore: %d lattice arcs -> product %zu states / %zu arcs, %zu + %zu determinised grammar states, %zu + %zu memo entries, built in %.1f ms\n",
                      n_arcs, pst.size(), psrc.size(), Lo.states.size(), Ln.states.size(), Lo.memo_size(), Ln.memo_size(),
                      std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_in).count());
  return b2t_lattice_nbest_core((int)pst.size(), 0, (int)psrc.size(), psrc.data(), pdst.data(), pil.data(), pol.data(), pg.data(), pa.data(),
                                pd.data(), (int)fst_.size(), fst_.data(), fc.data(), fd.data(), nbest, beam, out_words, w_off, w_cap,
                                out_ali, a_off, a_cap, costs);
}
