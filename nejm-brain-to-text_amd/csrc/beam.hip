// beam.hip — batched CTC prefix beam search (LM-free searcher of the reference's decoder:
// language_model/runtime/core/decoder/ctc_prefix_beam_search.cc:44-136, PrefixScore at
// ctc_prefix_beam_search.h:27-42, LogAdd at language_model/runtime/core/utils/utils.cc:24-30).
//
// One workgroup per utterance; the live beam (<= 128 prefixes: scores, Viterbi scores, trie node ids) lives
// in LDS for the whole call, frames are consumed sequentially.  Prefixes are nodes of a per-utterance trie in
// HBM (parent/token/depth arrays + an open-addressing hash on (parent, token)), so prefix identity — the
// std::unordered_map<vector<int>> key of the reference — is a node id and merging two ways of reaching the
// same prefix is an integer compare.  Every frame is evaluated in GATHER form: one thread per output
// candidate ("prefix h stays" / "prefix h extended by top-k class c") collects the <= 3 contributions the
// reference's scatter loops would add to it (blank, repeated last token, extension of the parent prefix), so
// the log-sum order is fixed and the result is deterministic.  Survivors are ranked by counting and written
// in sorted order.  Viterbi token times are kept per prefix like the reference (times_s / times_ns).
// State persists in HBM between calls: Search() can be fed frame by frame (streaming decode).
//
// Optional n-gram fusion (b2t_prefix_beam_search_lm_f32): a token-level back-off LM, resident in HBM as an automaton
// (ngram_lm.py: dense child table per n-gram node, ln p, ln back-off, suffix link, next state), adds
// alpha * ln p(token | history) + beta per emitted token.  The LM score and LM state depend on the prefix only, so they
// are stored once per trie node; the second beam is pruned, and hypotheses are ranked, by CTC score + LM score.  The
// reference reaches its LMs through a word-level WFST that cannot be built or pinned here: this part is
// pinned by the oracle's restatement only.
#include "common.h"

namespace b2t {

constexpr int BMAX = 128;          // max second_beam_size (the candidate arrays are dynamic LDS: 60 B x beam x (first_beam + 1))
constexpr int KMAX = 16;           // max first_beam_size
constexpr float NEGMAX = -3.402823466e38f;   // -FLT_MAX: the reference's log-zero sentinel

__device__ __forceinline__ float log_add(float x, float y) {
  if (x <= NEGMAX) return y;
  if (y <= NEGMAX) return x;
  const float m = fmaxf(x, y);
  return logf(expf(x - m) + expf(y - m)) + m;
}

// Back-off automaton of the fused n-gram LM (all tables in HBM; child == nullptr: no LM).
struct LmArgs {
  const int* child; const float* logp; const float* bow; const int* suffix; const int* nstate;
  int V, start, eos;
  float alpha, beta, unk;
  // word-level mode (lex_child != nullptr): pronunciation trie + sparse children of the word LM; `child` is unused
  const int* lex_child; const int* lex_wbeg; const int* lex_wend; const int* wlist;
  const int* cb; const int* ce; const int* ctok; const int* cnode;
  int sil;
};

// the same with sorted child arrays (large vocabularies): bisection in ctok[cb[s] .. ce[s])
__device__ __forceinline__ float lm_step_sparse(const LmArgs& lm, int state, int w, int* next) {
  float acc = 0.f;
  int s = state;
  for (int it = 0; it < 16; ++it) {
    int lo = lm.cb[s], hi = lm.ce[s];
    const int end = hi;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (lm.ctok[mid] < w) lo = mid + 1; else hi = mid;
    }
    if (lo < end && lm.ctok[lo] == w) { const int c = lm.cnode[lo]; *next = lm.nstate[c]; return acc + lm.logp[c]; }
    if (s == 0) break;
    acc += lm.bow[s];
    s = lm.suffix[s];
  }
  *next = 0;
  return acc + lm.unk;
}

// Word emission at lexicon node lx (a word end): the homophone the LM likes best (ties -> lowest word id).
__device__ __forceinline__ float emit_word(const LmArgs& lm, int lx, int state, int* next) {
  float best = -INFINITY; int bn = 0;
  for (int k = lm.lex_wbeg[lx]; k < lm.lex_wend[lx]; ++k) {
    int ns;
    const float lp = lm_step_sparse(lm, state, lm.wlist[k], &ns);
    if (lp > best) { best = lp; bn = ns; }
  }
  *next = bn;
  return best;
}

// ln p(w | state) and the state after emitting w: at most `order` dependent steps.
__device__ __forceinline__ float lm_step(const LmArgs& lm, int state, int w, int* next) {
  float acc = 0.f;
  int s = state;
  for (int it = 0; it < 16; ++it) {
    const int c = lm.child[(long long)s * lm.V + w];
    if (c >= 0) { *next = lm.nstate[c]; return acc + lm.logp[c]; }
    if (s == 0) break;
    acc += lm.bow[s];
    s = lm.suffix[s];
  }
  *next = 0;
  return acc + lm.unk;
}

// ---- state layout (ints unless noted), per utterance ------------------------------------------------
//   hdr[8]: nb, abs_t, node_count, cur, overflow
//   parent[NN] token[NN] depth[NN] lm[NN] (float) lmstate[NN]  hkey[HT] (u64)  hval[HT]
//   hyp buffers x2: node[BMAX], fl[5][BMAX] (s, ns, v_s, v_ns, ctp), times_s[BMAX][L], times_ns[BMAX][L]
struct BeamLayout {
  int NN, HT, L;
  size_t o_parent, o_token, o_depth, o_nlm, o_nlst, o_nlx, o_hkey, o_hval, o_hyp, hyp_stride, total;
  __host__ __device__ BeamLayout(int nn, int l) {
    NN = nn; L = l;
    HT = 1; while (HT < 2 * nn) HT <<= 1;
    size_t o = 8 * 4;
    o_parent = o; o += (size_t)NN * 4;
    o_token = o; o += (size_t)NN * 4;
    o_depth = o; o += (size_t)NN * 4;
    o_nlm = o; o += (size_t)NN * 4;     // LM score of the prefix (float)
    o_nlst = o; o += (size_t)NN * 4;    // LM state of the prefix
    o_nlx = o; o += (size_t)NN * 4;     // lexicon-trie node of the prefix (word-level mode)
    o = (o + 7) & ~(size_t)7;
    o_hkey = o; o += (size_t)HT * 8;
    o_hval = o; o += (size_t)HT * 4;
    o_hyp = o;
    hyp_stride = ((size_t)BMAX * 4 + (size_t)5 * BMAX * 4 + (size_t)2 * BMAX * L * 4);
    o += 2 * hyp_stride;
    total = (o + 255) & ~(size_t)255;
  }
};

struct HypBuf {
  int* node; float* fl; int* ts; int* tns;
  __device__ HypBuf(unsigned char* base, const BeamLayout& lay, int which) {
    unsigned char* p = base + lay.o_hyp + (size_t)which * lay.hyp_stride;
    node = reinterpret_cast<int*>(p); p += BMAX * 4;
    fl = reinterpret_cast<float*>(p); p += 5 * BMAX * 4;
    ts = reinterpret_cast<int*>(p); p += (size_t)BMAX * lay.L * 4;
    tns = reinterpret_cast<int*>(p);
  }
};

__global__ void beam_reset_kernel(unsigned char* state, size_t per_utt, int NN, int L) {
  BeamLayout lay(NN, L);
  unsigned char* st = state + (size_t)blockIdx.x * per_utt;
  int* hdr = reinterpret_cast<int*>(st);
  unsigned long long* hkey = reinterpret_cast<unsigned long long*>(st + lay.o_hkey);
  int* hval0 = reinterpret_cast<int*>(st + lay.o_hval);
  for (int i = threadIdx.x; i < lay.HT; i += blockDim.x) { hkey[i] = ~0ull; hval0[i] = -1; }
  if (threadIdx.x == 0) {
    hdr[0] = 1; hdr[1] = 0; hdr[2] = 1; hdr[3] = 0; hdr[4] = 0;
    reinterpret_cast<int*>(st + lay.o_parent)[0] = -1;   // node 0 = empty prefix
    reinterpret_cast<int*>(st + lay.o_token)[0] = -1;
    reinterpret_cast<int*>(st + lay.o_depth)[0] = 0;
    reinterpret_cast<float*>(st + lay.o_nlm)[0] = 0.f;
    reinterpret_cast<int*>(st + lay.o_nlst)[0] = -1;   // LM state of the empty prefix: set by the first fused search
    reinterpret_cast<int*>(st + lay.o_nlx)[0] = 0;
    HypBuf hb(st, lay, 0);
    hb.node[0] = 0;
    hb.fl[0 * BMAX] = 0.f;      // s
    hb.fl[1 * BMAX] = NEGMAX;   // ns
    hb.fl[2 * BMAX] = 0.f;      // v_s
    hb.fl[3 * BMAX] = 0.f;      // v_ns   (ctc_prefix_beam_search.cc:27-31)
    hb.fl[4 * BMAX] = NEGMAX;   // cur_token_prob
  }
}

__device__ int trie_find_or_add(int parent, int token, int* par, int* tok, int* dep, unsigned long long* hkey,
                                int* hval, int HT, int NN, int* hdr) {
  const unsigned long long key = ((unsigned long long)(unsigned)parent << 32) | (unsigned)token;
  unsigned h = (unsigned)((key * 0x9E3779B97F4A7C15ull) >> 40) & (unsigned)(HT - 1);
  for (int probe = 0; probe < HT; ++probe) {
    unsigned long long k = hkey[h];
    if (k == key) {
      int v;
      while ((v = __hip_atomic_load(&hval[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < 0) {}
      return v;
    }
    if (k == ~0ull) {
      const unsigned long long prev = atomicCAS(&hkey[h], ~0ull, key);
      if (prev == ~0ull) {
        const int id = atomicAdd(&hdr[2], 1);
        if (id >= NN) { hdr[4] = 1; hval[h] = 0; return 0; }   // overflow flag; results invalid
        par[id] = parent; tok[id] = token; dep[id] = dep[parent] + 1;
        __hip_atomic_store(&hval[h], id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return id;
      }
      if (prev == key) continue;   // someone else inserted the same key: re-read this slot
    }
    h = (h + 1) & (unsigned)(HT - 1);
  }
  hdr[4] = 1;
  return 0;
}

// times-vector source of a candidate: hyp index, which vector (0 = times_s, 1 = times_ns), op
// (0 copy, 1 copy + append t, 2 copy + overwrite last with t, 3 empty)
struct TSrc { short h; char vec; char op; };

// rank[ci] = number of keys larger than key[ci] (0 keys = invalid candidates get 1 << 20).  The workgroup is split into
// blockDim / 256 parts: a thread keeps NP candidates' keys in registers (candidates lane, lane + 256, ...) and streams over ITS
// part's share of the key array, two keys per LDS read; the partial counts meet in LDS atomics.
template <int NP>
__device__ __forceinline__ void rank_by_counting(const unsigned long long* __restrict__ key, int* __restrict__ rank, int n) {
  const int lane = (int)threadIdx.x & 255, part = (int)threadIdx.x >> 8, nparts = ((int)blockDim.x + 255) >> 8;
  for (int ci = (int)threadIdx.x; ci < n; ci += (int)blockDim.x) rank[ci] = key[ci] != 0ull ? 0 : (1 << 20);
  __syncthreads();
  unsigned long long mine[NP];
  int r[NP];
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    const int ci = lane + j * 256;
    mine[j] = ci < n ? key[ci] : ~0ull;     // nothing is larger than the padding
    r[j] = 0;
  }
  const int pairs = n >> 1, per = (pairs + nparts - 1) / nparts;
  const int p0 = part * per, p1 = min(pairs, p0 + per);
#pragma unroll 2
  for (int x = p0; x < p1; ++x) {
    const ulonglong2 k = *reinterpret_cast<const ulonglong2*>(key + 2 * x);     // 16-byte aligned: the key array starts the dynamic LDS
#pragma unroll
    for (int j = 0; j < NP; ++j) r[j] += (k.x > mine[j]) + (k.y > mine[j]);
  }
  if ((n & 1) && part == nparts - 1) {
    const unsigned long long k = key[n - 1];
#pragma unroll
    for (int j = 0; j < NP; ++j) r[j] += (k > mine[j]);
  }
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    const int ci = lane + j * 256;
    if (ci < n && mine[j] != 0ull && r[j]) atomicAdd(&rank[ci], r[j]);
  }
}

#ifdef B2T_BEAM_TIMING
#define BT(i) { if (threadIdx.x == 0) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); bt_acc[i] += now_ - bt_prev; bt_prev = now_; } }
#else
#define BT(i)
#endif
__global__ __launch_bounds__(1024) void prefix_beam_kernel(const float* __restrict__ logp, const int32_t* __restrict__ lens,
                                                          int T, int C, int K, int beam, int blank,
                                                          unsigned char* state, size_t per_utt, int NN, int L,
                                                          int32_t* __restrict__ hyps, int32_t* __restrict__ hyp_len,
                                                          float* __restrict__ score, float* __restrict__ vscore,
                                                          int32_t* __restrict__ times, LmArgs lm,
                                                          float* __restrict__ lm_score, int ncmax) {
  const int u = blockIdx.x, tid = threadIdx.x;
  const bool lexm = lm.lex_child != nullptr;
  const bool fused = lm.child != nullptr || lexm;
  BeamLayout lay(NN, L);
  unsigned char* st = state + (size_t)u * per_utt;
  int* hdr = reinterpret_cast<int*>(st);
  int* par = reinterpret_cast<int*>(st + lay.o_parent);
  int* tok = reinterpret_cast<int*>(st + lay.o_token);
  int* dep = reinterpret_cast<int*>(st + lay.o_depth);
  float* nlm = reinterpret_cast<float*>(st + lay.o_nlm);
  int* nlst = reinterpret_cast<int*>(st + lay.o_nlst);
  int* nlx = reinterpret_cast<int*>(st + lay.o_nlx);
  unsigned long long* hkey = reinterpret_cast<unsigned long long*>(st + lay.o_hkey);
  int* hval = reinterpret_cast<int*>(st + lay.o_hval);

  // candidate arrays: ncmax = beam * (K + 1) entries each, dynamic LDS
  extern __shared__ __attribute__((aligned(16))) unsigned char cand_raw[];
  // ranking key of a candidate: (order-preserving bits of its score) << 32 | ~index; 0 = invalid.  "How many keys
  // are larger than mine" is then one 64-bit LDS read and one compare per candidate (score descending, index ascending)
  unsigned long long* c_key = reinterpret_cast<unsigned long long*>(cand_raw);
  float* c_lm = reinterpret_cast<float*>(c_key + ncmax);
  int* c_lst = reinterpret_cast<int*>(c_lm + ncmax);
  int* c_lx = c_lst + ncmax;
  int* c_valid = c_lx + ncmax;
  int* c_node = c_valid + ncmax;
  int* c_tok = c_node + ncmax;
  int* c_rank = c_tok + ncmax;
  float* c_s = reinterpret_cast<float*>(c_rank + ncmax);
  float* c_ns = c_s + ncmax;
  float* c_vs = c_ns + ncmax;
  float* c_vns = c_vs + ncmax;
  float* c_ctp = c_vns + ncmax;
  TSrc* c_ts = reinterpret_cast<TSrc*>(c_ctp + ncmax);
  TSrc* c_tn = c_ts + ncmax;
  __shared__ float h_lm[BMAX];
  __shared__ int h_lst[BMAX], h_lx[BMAX];
  __shared__ int h_node[BMAX], h_par[BMAX], h_tok[BMAX], h_dep[BMAX];
  __shared__ float h_s[BMAX], h_ns[BMAX], h_vs[BMAX], h_vns[BMAX], h_ctp[BMAX], h_score[BMAX], h_vit[BMAX];
  __shared__ int tk_id[KMAX]; __shared__ float tk_p[KMAX];
  __shared__ float cp[64];        // class log-probs of the frame; selection marks
  __shared__ int s_nb, s_cur, s_abs, s_nvalid;
  __shared__ int s_rk2ci[BMAX];
  __shared__ int h_pidx[BMAX];                    // beam index of a hypothesis' parent prefix (-1: the parent is not in the beam)
  __shared__ unsigned long long h_cmask[BMAX];    // tokens c for which prefix h + c is itself in the beam (C <= 64)

  if (tid == 0) {
    s_nb = hdr[0]; s_abs = hdr[1]; s_cur = hdr[3];
    if (fused && nlst[0] < 0) nlst[0] = lm.start;   // first fused search after a reset
  }
  __syncthreads();
  {
    HypBuf hb(st, lay, s_cur);
    if (tid < s_nb) {
      const int n = hb.node[tid];
      h_lm[tid] = fused ? nlm[n] : 0.f; h_lst[tid] = fused ? nlst[n] : 0; h_lx[tid] = lexm ? nlx[n] : 0;
      h_node[tid] = n; h_par[tid] = par[n]; h_tok[tid] = tok[n]; h_dep[tid] = dep[n];
      h_s[tid] = hb.fl[0 * BMAX + tid]; h_ns[tid] = hb.fl[1 * BMAX + tid]; h_vs[tid] = hb.fl[2 * BMAX + tid];
      h_vns[tid] = hb.fl[3 * BMAX + tid]; h_ctp[tid] = hb.fl[4 * BMAX + tid];
    }
  }
  __syncthreads();

  int Tu = lens ? lens[u] : T;
  if (Tu > T) Tu = T;
  const float* lp_u = logp + (long long)u * T * C;

#ifdef B2T_BEAM_TIMING
  unsigned long long bt_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, bt_prev = __builtin_amdgcn_s_memtime();
#endif
  for (int f = 0; f < Tu; ++f) {
    const int nb = s_nb, cur = s_cur, at = s_abs;
    BT(0)
    // ---- 1. first beam: top-K classes of the frame (one wave; ties -> lowest class id) --------------
    if (tid < 64) {
      float v = tid < C ? lp_u[(long long)f * C + tid] : -INFINITY;
      cp[tid] = v;
      for (int k = 0; k < K; ++k) {
        float best = v; int bi = tid;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
          const float ob = __shfl_xor(best, o); const int oi = __shfl_xor(bi, o);
          if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        if (tid == 0) { tk_id[k] = bi; tk_p[k] = best; }
        if (tid == bi) v = -INFINITY;
      }
    }
    if (tid < nb) {
      h_score[tid] = log_add(h_s[tid], h_ns[tid]);
      h_vit[tid] = h_vs[tid] > h_vns[tid] ? h_vs[tid] : h_vns[tid];
      h_cmask[tid] = 0ull;
    }
    __syncthreads();
    // Who is whose parent, once per frame (trie nodes are unique in the beam): every candidate used to search the beam for
    // its parent / for an existing child -- 100 LDS reads each, 110 k per beam-100 frame.
    if (tid < nb) {
      int hp = -1;
      const int want = h_par[tid];
      for (int x = 0; x < nb; ++x) if (h_node[x] == want) hp = x;
      h_pidx[tid] = hp;
      if (hp >= 0 && h_dep[tid] > 0 && h_tok[tid] >= 0 && h_tok[tid] < 64) atomicOr(&h_cmask[hp], 1ull << h_tok[tid]);
    }
    __syncthreads();

    BT(1)   // top-K classes
    // ---- 2. candidates (gather form) --------------------------------------------------------------------
    const int ncand = nb * (K + 1);
    for (int ci = tid; ci < ncand; ci += blockDim.x) {
      const int h = ci / (K + 1), slot = ci - h * (K + 1);
      float ns_ = NEGMAX, s_ = NEGMAX, vs_ = NEGMAX, vns_ = NEGMAX, ctp_ = NEGMAX;
      TSrc ts{0, 0, 3}, tn{0, 0, 3};
      int valid = 0, node = -1, token = -1, pnode = -1;
      float lmv = h_lm[h]; int lst = h_lst[h], lxv = h_lx[h];   // a prefix that stays keeps its LM score and states
      const int hvec = h_vs[h] > h_vns[h] ? 0 : 1;     // which vector PrefixScore::times() returns
      if (slot == 0) {                                   // prefix h stays
        node = h_node[h];
        const int last = h_tok[h];
        for (int k = 0; k < K; ++k) {
          const int c = tk_id[k]; const float p = tk_p[k];
          if (c == blank) {                              // case 0: *a + blank => *a
            s_ = log_add(s_, h_score[h] + p);
            vs_ = h_vit[h] + p;
            ts = TSrc{(short)h, (char)hvec, 0};
            valid = 1;
          } else if (h_dep[h] > 0 && c == last) {        // case 1: *a + a => *a
            ns_ = log_add(ns_, h_ns[h] + p);
            if (vns_ < h_vns[h] + p) {
              vns_ = h_vns[h] + p;
              if (ctp_ < p) { ctp_ = p; tn = TSrc{(short)h, 1, 2}; }
            }
            valid = 1;
          }
        }
        if (h_dep[h] > 0) {                              // extension of the parent prefix that lands on h
          const int hp = h_pidx[h];
          if (hp >= 0) {
            for (int k = 0; k < K; ++k) {
              if (tk_id[k] != last || last == blank) continue;
              const float p = tk_p[k];
              float add, vc; TSrc src;
              if (h_dep[hp] > 0 && last == h_tok[hp]) {  // case 2: *a(blank) + a => *aa
                add = h_s[hp] + p; vc = h_vs[hp] + p; src = TSrc{(short)hp, 0, 1};
              } else {                                   // case 3: *a + b => *ab
                add = h_score[hp] + p; vc = h_vit[hp] + p;
                src = TSrc{(short)hp, (char)(h_vs[hp] > h_vns[hp] ? 0 : 1), 1};
              }
              ns_ = log_add(ns_, add);
              if (vns_ < vc) { vns_ = vc; ctp_ = p; tn = src; }
              valid = 1;
            }
          }
        }
      } else {                                           // prefix h extended by class c (new prefix)
        const int c = tk_id[slot - 1]; const float p = tk_p[slot - 1];
        if (c != blank) {
          const bool merged = (h_cmask[h] >> c) & 1ull;   // already a live prefix: gathered by its "stay" slot
          // word-level mode: the extension must follow the pronunciation trie; SIL closes a word (or is free at the root)
          bool ok = true;
          if (lexm && !merged) {
            const int lx = h_lx[h];
            if (c == lm.sil) {
              if (lx != 0) {
                if (lm.lex_wend[lx] > lm.lex_wbeg[lx]) {
                  int ns2;
                  lmv = h_lm[h] + lm.alpha * emit_word(lm, lx, h_lst[h], &ns2) + lm.beta;
                  lst = ns2;
                } else {
                  ok = false;
                }
              }
              lxv = 0;
            } else {
              lxv = lm.lex_child[(long long)lx * C + c];
              ok = lxv >= 0;
            }
          }
          if (!merged && ok) {
            float add, vc; TSrc src;
            if (h_dep[h] > 0 && c == h_tok[h]) { add = h_s[h] + p; vc = h_vs[h] + p; src = TSrc{(short)h, 0, 1}; }
            else { add = h_score[h] + p; vc = h_vit[h] + p; src = TSrc{(short)h, (char)hvec, 1}; }
            ns_ = log_add(ns_, add);
            if (vns_ < vc) { vns_ = vc; ctp_ = p; tn = src; }
            valid = 1; pnode = h_node[h]; token = c;
            if (fused && !lexm) {   // LM score / state of the NEW prefix (a function of the prefix only)
              int ns2;
              lmv = h_lm[h] + lm.alpha * lm_step(lm, h_lst[h], c, &ns2) + lm.beta;
              lst = ns2;
            }
          }
        }
      }
      c_valid[ci] = valid; c_node[ci] = slot == 0 ? node : -1 - pnode; c_tok[ci] = token;
      c_s[ci] = s_; c_ns[ci] = ns_; c_vs[ci] = vs_; c_vns[ci] = vns_; c_ctp[ci] = ctp_;
      c_lm[ci] = lmv; c_lst[ci] = lst; c_lx[ci] = lxv;
      {
        const float scv = log_add(s_, ns_) + (fused ? lmv : 0.f);
        unsigned bits = __float_as_uint(scv);
        bits = (bits & 0x80000000u) ? ~bits : (bits | 0x80000000u);   // monotone float -> unsigned
        c_key[ci] = valid ? (((unsigned long long)bits << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)ci)) : 0ull;
      }
      c_ts[ci] = ts; c_tn[ci] = tn;
    }
    __syncthreads();

    BT(2)   // candidates (+ LM)
    // ---- 3. second beam: rank by counting (ties -> lower candidate index), keep the best `beam` ---------
    // (every thread keeps its candidates' keys in registers and streams over a share of the key array once, two keys per LDS
    // read: with one pass over the whole array per candidate this phase was 47 % of a beam-100 frame)
    switch ((ncand + 255) / 256) {
      case 0: break;
      case 1: rank_by_counting<1>(c_key, c_rank, ncand); break;
      case 2: rank_by_counting<2>(c_key, c_rank, ncand); break;
      case 3: rank_by_counting<3>(c_key, c_rank, ncand); break;
      case 4: rank_by_counting<4>(c_key, c_rank, ncand); break;
      case 5: rank_by_counting<5>(c_key, c_rank, ncand); break;
      case 6: rank_by_counting<6>(c_key, c_rank, ncand); break;
      default: rank_by_counting<9>(c_key, c_rank, ncand); break;     // ncmax <= 128 * 17 = 2176 <= 9 * 256 candidates
    }
    __syncthreads();

    BT(3)   // ranking
    // ---- 4. write survivors (sorted) into the other hypothesis buffer ------------------------------------
    HypBuf hc(st, lay, cur), hn(st, lay, cur ^ 1);
    if (tid < BMAX) s_rk2ci[tid] = -1;
    if (tid == 0) s_nvalid = 0;
    __syncthreads();
    {   // number of valid candidates (one LDS atomic per wave)
      int mine = 0;
      for (int ci = tid; ci < ncand; ci += blockDim.x) mine += c_valid[ci];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o);
      if ((tid & 63) == 0 && mine) atomicAdd(&s_nvalid, mine);
    }
    for (int ci = tid; ci < ncand; ci += blockDim.x) {
      const int r = c_rank[ci];
      if (r >= beam) continue;
      int node = c_node[ci];
      if (node < 0) {
        node = trie_find_or_add(-1 - node, c_tok[ci], par, tok, dep, hkey, hval, lay.HT, lay.NN, hdr);
        if (fused) { nlm[node] = c_lm[ci]; nlst[node] = c_lst[ci]; if (lexm) nlx[node] = c_lx[ci]; }   // same values if it existed
      }
      hn.node[r] = node;
      hn.fl[0 * BMAX + r] = c_s[ci]; hn.fl[1 * BMAX + r] = c_ns[ci]; hn.fl[2 * BMAX + r] = c_vs[ci];
      hn.fl[3 * BMAX + r] = c_vns[ci]; hn.fl[4 * BMAX + r] = c_ctp[ci];
      s_rk2ci[r] = ci;
    }
    __syncthreads();
    BT(4)   // survivors: trie + scores
    // token-time vectors of the survivors: all threads share the copies (one element each) -- a survivor copying its
    // own two vectors serially was ~80 % of a frame's time (2 x depth dependent HBM round trips)
    {
      const int nsurv = min(beam, ncand);
      const int Lc = min(lay.L, at + 2);   // a prefix is never longer than the number of frames seen
      for (int idx = tid; idx < nsurv * 2 * Lc; idx += blockDim.x) {
        const int i = idx % Lc, which = (idx / Lc) & 1, r = idx / (2 * Lc);
        const int ci = s_rk2ci[r];
        if (ci < 0) continue;
        const TSrc src = which == 0 ? c_ts[ci] : c_tn[ci];
        if (src.op == 3) continue;
        const int n = h_dep[src.h];
        int* dst = (which == 0 ? hn.ts : hn.tns) + (size_t)r * lay.L;
        const int* sv = (src.vec == 0 ? hc.ts : hc.tns) + (size_t)src.h * lay.L;
        if (src.op == 1 && i == n) dst[i] = at;
        else if (src.op == 2 && i == n - 1) dst[i] = at;
        else if (i < n) dst[i] = sv[i];
      }
    }
    __syncthreads();
    BT(5)   // time vectors
    if (tid == 0) {
      const int cnt = s_nvalid;
      s_nb = cnt < beam ? cnt : beam; s_cur = cur ^ 1; s_abs = at + 1;
    }
    __threadfence_block();
    __syncthreads();
    // reload the new beam into LDS (node ids were assigned by other threads: read them back from the buffer)
    {
      HypBuf hb(st, lay, s_cur);
      if (tid < s_nb) {
        const int n = hb.node[tid];
        h_lm[tid] = fused ? nlm[n] : 0.f; h_lst[tid] = fused ? nlst[n] : 0; h_lx[tid] = lexm ? nlx[n] : 0;
        h_node[tid] = n; h_par[tid] = par[n]; h_tok[tid] = tok[n]; h_dep[tid] = dep[n];
        h_s[tid] = hb.fl[0 * BMAX + tid]; h_ns[tid] = hb.fl[1 * BMAX + tid]; h_vs[tid] = hb.fl[2 * BMAX + tid];
        h_vns[tid] = hb.fl[3 * BMAX + tid]; h_ctp[tid] = hb.fl[4 * BMAX + tid];
      }
    }
    __syncthreads();
  }

#ifdef B2T_BEAM_TIMING
  BT(6)
  if (threadIdx.x == 0 && blockIdx.x == 0)
    printf("beam u0 cycles: topk %llu | candidates %llu | ranking %llu | survivors %llu | times %llu | reload %llu (loop head %llu)\n", bt_acc[1], bt_acc[2],
           bt_acc[3], bt_acc[4], bt_acc[5], bt_acc[6], bt_acc[0]);
#endif
  // ---- results: hypotheses (token sequences by walking the trie), scores, Viterbi times ------------------
  if (tid == 0) { hdr[0] = s_nb; hdr[1] = s_abs; hdr[3] = s_cur; }
  {
    HypBuf hb(st, lay, s_cur);
    if (tid < beam) {
      const long long o = (long long)u * beam + tid;
      if (tid < s_nb) {
        const int n = h_dep[tid];
        hyp_len[o] = n;
        score[o] = log_add(h_s[tid], h_ns[tid]);
        vscore[o] = h_vs[tid] > h_vns[tid] ? h_vs[tid] : h_vns[tid];
        if (lm_score) {
          float l = h_lm[tid];
          int stt = h_lst[tid], dummy;
          if (lexm && h_lx[tid] != 0) {   // the last word has no closing SIL yet: emit it, or disqualify an unfinished word
            const int lx = h_lx[tid];
            if (lm.lex_wend[lx] > lm.lex_wbeg[lx]) l += lm.alpha * emit_word(lm, lx, stt, &stt) + lm.beta;
            else l = -INFINITY;
          }
          if (fused && lm.eos >= 0)
            l += lm.alpha * (lexm ? lm_step_sparse(lm, stt, lm.eos, &dummy) : lm_step(lm, stt, lm.eos, &dummy));
          lm_score[o] = l;
        }
        int node = h_node[tid];
        for (int i = n - 1; i >= 0; --i) { if (i < L) hyps[o * L + i] = tok[node]; node = par[node]; }
        const int* tv = (h_vs[tid] > h_vns[tid] ? hb.ts : hb.tns) + (size_t)tid * lay.L;
        if (times) for (int i = 0; i < n && i < L; ++i) times[o * L + i] = tv[i];
      } else {
        hyp_len[o] = -1; score[o] = NEGMAX; vscore[o] = NEGMAX;
        if (lm_score) lm_score[o] = 0.f;
      }
    }
  }
}

}  // namespace b2t

using namespace b2t;

// candidate slots of a frame and the dynamic LDS they take (one 8-byte and 14 4-byte arrays)
static int first_beam_ncmax(int first_beam, int second_beam) { return second_beam * (first_beam + 1); }
// Threads of a search workgroup (one per utterance).  Every phase of a frame is a strided loop over beam * (K + 1) candidates or
// over the survivors' time vectors, so wide beams take the full 1024 (beam 100: 0.80 -> 0.20 ms per utterance together with the
// single-pass ranking and the per-frame parent / child tables); a narrow beam has nothing to spread and keeps 256.
static int beam_threads(int second_beam) {
  static const int env = getenv("B2T_BEAM_THREADS") ? atoi(getenv("B2T_BEAM_THREADS")) : 0;
  if (env == 256 || env == 512 || env == 1024) return env;
  return second_beam >= 64 ? 1024 : (second_beam >= 24 ? 512 : 256);
}
static size_t beam_cand_bytes(int first_beam, int second_beam) {   // <= 128 * 17 * 64 B = 139 KB of the CU's 160 KB
  const size_t b = (size_t)first_beam_ncmax(first_beam, second_beam) * 16 * 4;
  static size_t raised = 0;
  if (b > 48 * 1024 && b > raised) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(prefix_beam_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)b);
    raised = b;
  }
  return b;
}

extern "C" size_t b2t_beam_state_bytes(int max_len, int max_nodes) {
  BeamLayout lay(max_nodes, max_len);
  return lay.total;
}

extern "C" int b2t_beam_reset(void* state, int U, int max_len, int max_nodes, void* stream) {
  B2T_REQUIRE(state && U > 0 && max_len > 0 && max_nodes > 1, "beam_reset: bad args");
  BeamLayout lay(max_nodes, max_len);
  hipLaunchKernelGGL(beam_reset_kernel, dim3(U), dim3(256), 0, as_stream(stream), reinterpret_cast<unsigned char*>(state),
                     lay.total, max_nodes, max_len);
  B2T_CHECK_LAUNCH("b2t_beam_reset");
  return 0;
}

extern "C" int b2t_prefix_beam_search_f32(const float* logp, const int32_t* lens, int U, int T, int C, int first_beam,
                                          int second_beam, int blank, void* state, int max_len, int max_nodes,
                                          int32_t* hyps, int32_t* hyp_len, float* score, float* vscore, int32_t* times,
                                          void* stream) {
  B2T_REQUIRE(logp && state && U > 0 && T > 0 && C > 1 && C <= 64, "prefix_beam_search: bad shape U=%d T=%d C=%d", U, T, C);
  B2T_REQUIRE(first_beam >= 1 && second_beam >= 1 && second_beam <= BMAX, "prefix_beam_search: beams out of range (<=%d)", BMAX);
  if (first_beam > C) first_beam = C;
  B2T_REQUIRE(first_beam <= KMAX, "prefix_beam_search: first_beam_size <= %d", KMAX);
  BeamLayout lay(max_nodes, max_len);
  hipLaunchKernelGGL(prefix_beam_kernel, dim3(U), dim3(beam_threads(second_beam)), beam_cand_bytes(first_beam, second_beam), as_stream(stream), logp, lens, T, C, first_beam, second_beam,
                     blank, reinterpret_cast<unsigned char*>(state), lay.total, max_nodes, max_len, hyps, hyp_len, score,
                     vscore, times, LmArgs{nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, -1, 0.f, 0.f, 0.f, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0},
                     static_cast<float*>(nullptr), first_beam_ncmax(first_beam, second_beam));
  B2T_CHECK_LAUNCH("b2t_prefix_beam_search_f32");
  return 0;
}

extern "C" int b2t_prefix_beam_search_lm_f32(const float* logp, const int32_t* lens, int U, int T, int C, int first_beam,
                                             int second_beam, int blank, void* state, int max_len, int max_nodes,
                                             int32_t* hyps, int32_t* hyp_len, float* score, float* vscore, int32_t* times,
                                             const int32_t* lm_child, const float* lm_logp, const float* lm_bow,
                                             const int32_t* lm_suffix, const int32_t* lm_nstate, int lm_vocab,
                                             int lm_start_state, int lm_eos, float alpha, float beta, float unk_logp,
                                             float* lm_score, void* stream) {
  B2T_REQUIRE(logp && state && U > 0 && T > 0 && C > 1 && C <= 64, "prefix_beam_search_lm: bad shape U=%d T=%d C=%d", U, T, C);
  B2T_REQUIRE(first_beam >= 1 && second_beam >= 1 && second_beam <= BMAX, "prefix_beam_search_lm: beams out of range (<=%d)", BMAX);
  B2T_REQUIRE(lm_child && lm_logp && lm_bow && lm_suffix && lm_nstate && lm_vocab >= C, "prefix_beam_search_lm: LM tables missing");
  B2T_REQUIRE(lm_eos < lm_vocab && lm_start_state >= 0, "prefix_beam_search_lm: bad LM vocabulary / start state");
  if (first_beam > C) first_beam = C;
  B2T_REQUIRE(first_beam <= KMAX, "prefix_beam_search_lm: first_beam_size <= %d", KMAX);
  BeamLayout lay(max_nodes, max_len);
  hipLaunchKernelGGL(prefix_beam_kernel, dim3(U), dim3(beam_threads(second_beam)), beam_cand_bytes(first_beam, second_beam), as_stream(stream), logp, lens, T, C, first_beam, second_beam,
                     blank, reinterpret_cast<unsigned char*>(state), lay.total, max_nodes, max_len, hyps, hyp_len, score,
                     vscore, times,
                     LmArgs{lm_child, lm_logp, lm_bow, lm_suffix, lm_nstate, lm_vocab, lm_start_state, lm_eos, alpha, beta, unk_logp, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0},
                     lm_score, first_beam_ncmax(first_beam, second_beam));
  B2T_CHECK_LAUNCH("b2t_prefix_beam_search_lm_f32");
  return 0;
}

extern "C" int b2t_prefix_beam_search_lex_f32(const float* logp, const int32_t* lens, int U, int T, int C, int first_beam,
                                              int second_beam, int blank, void* state, int max_len, int max_nodes,
                                              int32_t* hyps, int32_t* hyp_len, float* score, float* vscore, int32_t* times,
                                              const b2t_lexlm_t* d, float* lm_score, void* stream) {
  B2T_REQUIRE(logp && state && d && U > 0 && T > 0 && C > 1 && C <= 64, "prefix_beam_search_lex: bad shape U=%d T=%d C=%d", U, T, C);
  B2T_REQUIRE(first_beam >= 1 && second_beam >= 1 && second_beam <= BMAX, "prefix_beam_search_lex: beams out of range (<=%d)", BMAX);
  B2T_REQUIRE(d->lex_child && d->lex_wbeg && d->lex_wend && d->wlist && d->lm_cb && d->lm_ce && d->lm_ctok && d->lm_cnode &&
                  d->lm_logp && d->lm_bow && d->lm_suffix && d->lm_nstate, "prefix_beam_search_lex: lexicon / LM tables missing");
  B2T_REQUIRE(d->sil > 0 && d->sil < C && d->sil != blank, "prefix_beam_search_lex: sil class %d out of range", d->sil);
  if (first_beam > C) first_beam = C;
  B2T_REQUIRE(first_beam <= KMAX, "prefix_beam_search_lex: first_beam_size <= %d", KMAX);
  BeamLayout lay(max_nodes, max_len);
  LmArgs lm{nullptr, d->lm_logp, d->lm_bow, d->lm_suffix, d->lm_nstate, 0, d->lm_start_state, d->lm_eos, d->alpha, d->beta,
            d->unk_logp, d->lex_child, d->lex_wbeg, d->lex_wend, d->wlist, d->lm_cb, d->lm_ce, d->lm_ctok, d->lm_cnode, d->sil};
  hipLaunchKernelGGL(prefix_beam_kernel, dim3(U), dim3(beam_threads(second_beam)), beam_cand_bytes(first_beam, second_beam), as_stream(stream), logp, lens, T, C, first_beam, second_beam,
                     blank, reinterpret_cast<unsigned char*>(state), lay.total, max_nodes, max_len, hyps, hyp_len, score,
                     vscore, times, lm, lm_score, first_beam_ncmax(first_beam, second_beam));
  B2T_CHECK_LAUNCH("b2t_prefix_beam_search_lex_f32");
  return 0;
}

// 1 if any utterance's trie overflowed max_nodes / max_len (results invalid); synchronises the stream.
extern "C" int b2t_beam_overflowed(const void* state, int U, int max_len, int max_nodes, int* flag_host, void* stream) {
  B2T_REQUIRE(state && flag_host, "beam_overflowed: null");
  BeamLayout lay(max_nodes, max_len);
  *flag_host = 0;
  for (int u = 0; u < U; ++u) {
    int hdr[8];
    int rc = check_hip(hipMemcpyAsync(hdr, reinterpret_cast<const unsigned char*>(state) + (size_t)u * lay.total, sizeof(hdr),
                                      hipMemcpyDeviceToHost, as_stream(stream)), "beam_overflowed: copy");
    if (rc) return rc;
    rc = check_hip(hipStreamSynchronize(as_stream(stream)), "beam_overflowed: sync");
    if (rc) return rc;
    if (hdr[4]) *flag_host = 1;
  }
  return 0;
}
