// wfst.hip — batched WFST token passing for gfx950: the search inner loop of the reference's LM decoder
// (language_model/runtime/core/kaldi/decoder/lattice-faster-decoder.cc: ProcessEmitting :722-824, ProcessNonemitting
// :839-909, GetCutoff :650-720, FindOrAddToken :250-295, PruneForwardLinks(Final) :297-470) under the frame loop of
// CtcWfstBeamSearch::Search (language_model/runtime/core/decoder/ctc_wfst_beam_search.cc:70-121).
//
// Two searchers with identical results (tests/test_gpu_wfst.py::test_cluster_search_equals_single_workgroup):
//   * wfst_search_kernel: one workgroup per utterance; a frame's token hash (state -> token) lives in LDS when it fits
//     (<= 16384 slots = 128 KB);
//   * wfst_cluster_kernel (the default when the XCD round-robin probe passes): 8 workgroups per utterance placed on one
//     XCD, sharing the frame through that XCD's L2 (hash, work lists, counters in HBM-backed scratch read with L1-bypassing
//     loads), a monotonic L2 counter as the cluster barrier.
// The decode graph (T o L o G as CSR arcs: nejm-brain-to-text_amd/wfst.py, csrc/graphc.cpp) lives in HBM and is shared by all
// utterances; tokens and forward links of every frame are appended to the utterance's state block in HBM (they ARE the lattice).
//
// What is data-parallel here and sequential in the reference:
//   * ProcessEmitting tightens `next_cutoff` while it walks the token list, so which over-the-cutoff dead-end tokens get
//     created depends on hash-list order.  Here the frame's minimum candidate cost is reduced first and every candidate
//     below min + adaptive_beam is kept -- the reference's FINAL cutoff.  Tokens the reference creates beyond it are never
//     expanded (ProcessNonemitting and the next frame's cutoff skip them) and disappear in FinalizeDecoding, so the
//     pruned lattice, best path and n-best are the same.  (Only GetCutoff's max_active / min_active COUNTS can see such
//     tokens: a difference exists only while max_active binds in consecutive frames.)
//   * ProcessNonemitting's work queue becomes Bellman-Ford sweeps over the frame's tokens until no cost changes (the
//     cluster search: one pass in which the thread that lowers a token's cost relaxes that token's arcs itself); forward
//     links are generated once, after convergence, with the final costs (what the queue leaves behind).
//   * PruneActiveTokens every prune_interval frames is a memory optimisation (it only removes what FinalizeDecoding would
//     remove as well: its extra_costs are lower bounds), so it is a pass of its own between search calls (wfst_prune_kernel,
//     b2t_wfst_prune) that shares prune_frame() with b2t_wfst_finalize; n-best lists are bit-identical with and without it.
#include <float.h>
#include "common.h"

namespace b2t {
namespace {

constexpr int BP_NT = 256;   // threads of the best-path kernel (parallel argmin over the last frame; the backtrace: a serial chain walk + parallel gathers)
constexpr int BP_CAP = 1024;  // links of the best path handled per round of the backtrace
constexpr int NT = 1024;   // one workgroup per utterance; a frame holds thousands of tokens, each a dependent chain of gathers
constexpr unsigned UMAX = 0xffffffffu;
constexpr int MAX_C = 64;

struct Graph {
  const int* row; const int* ilabel; const int* olabel; const float* weight; const int* next; const int* n_eps;
  const float* final_cost; int start;
  // compact arcs (round 4; b2t_wfst_graph_t.compact): 10 bytes per arc instead of 16 -- labels = ilabel | olabel << 7 (one
  // word), the weight as IEEE half (|error| <= 2^-11 relative), next as before; the full-width arrays are then not read
  const unsigned* labels; const _Float16* w16; int compact;
};
__device__ __forceinline__ int g_il(const Graph& g, int a) { return g.compact ? (int)(g.labels[a] & 127u) : g.ilabel[a]; }
__device__ __forceinline__ int g_ol(const Graph& g, int a) { return g.compact ? (int)(g.labels[a] >> 7) : g.olabel[a]; }
__device__ __forceinline__ float g_w(const Graph& g, int a) { return g.compact ? (float)g.w16[a] : g.weight[a]; }
// the same with the arc format known at compile time (the cluster search's inner loops: the run-time test cost 3.5 %)
template <bool CP> __device__ __forceinline__ int g_il_t(const Graph& g, int a) { if constexpr (CP) return (int)(g.labels[a] & 127u); else return g.ilabel[a]; }
template <bool CP> __device__ __forceinline__ float g_w_t(const Graph& g, int a) { if constexpr (CP) return (float)g.w16[a]; else return g.weight[a]; }

// state block of one utterance (HBM), carved by layout(): header words then arrays
struct Hdr {
  int n_frames;        // decoded frames (emitting steps taken)
  int n_tok;           // tokens so far (all frames)
  int n_link;          // links so far
  int overflow;        // capacity exhausted (bit 0 tokens, 1 links, 2 hash slots, 3 frames): results invalid
  int num_input;       // input frames seen (incl. skipped ones)
  int is_last_blank, last_best;
  int finalized;
  float final_best;    // best (cost + final cost) on the last frame
  int has_final;
  unsigned arcs_lo, arcs_hi;   // emitting arcs expanded so far (64-bit): 16 B of graph each, the algorithmic traffic of the search
  int links_marked;    // links [0, links_marked) survived the last PruneActiveTokens pass (link_alive valid, all 1)
  int n_prunes;        // PruneActiveTokens passes so far
  int peak_tok, peak_link;   // high-water marks of n_tok / n_link (before the passes compacted them)
  int removed_tok, removed_link;   // what the PruneActiveTokens passes removed so far (created = held + removed)
};

// Scratch of the CLUSTER search (several workgroups per utterance, wfst_cluster_kernel below): every word is written with L2
// atomics or plain stores and read with L1-bypassing (sc1) loads by the workgroups of one cluster, which share an XCD's L2.
constexpr int WLG_CAP = 1 << 19;   // epsilon work list of a frame (tokens whose state has input-epsilon arcs; 125 k-word graphs put > 65 k of them into peak frames)
constexpr int HEAVY_CAP = 1 << 17; // heavy-token list of a frame: one 16-byte entry {token, its cost, first arc, end arc} per CHUNK of a heavy token's arcs (below)
constexpr int HEAVY_DEG = 32;
#ifndef B2T_CHASE_DEPTH
#define B2T_CHASE_DEPTH 4     // (-DB2T_CHASE_DEPTH=1 builds a library whose closure overflows all the time: the fallback rounds under test)
#endif
constexpr int CHASE_DEPTH = B2T_CHASE_DEPTH;    // tokens a thread of the epsilon closure may have pending (lowered, arcs not yet relaxed)
struct Clu {
  unsigned bar, bar_base; int pad0[62];          // cluster barrier: monotonic arrival counter, its value when the last launch ended
  // the counters the single-workgroup kernel keeps in LDS -- each on a 256-byte block of its own: they take ~2500 atomics per
  // frame between them (one per wave and trip), and atomics on words of one cache line are served one after the other
  // (all four in one line: 14.8 ms for the 32-utterance search; apart: 13.5)
  int n_tok, padt[63];
  int n_link, padl[63];
  int wl_n, padw[63];
  int overflow, pado[63];
  unsigned best[2], cand_min[2]; int narcs[2];   // per frame parity: cheapest token of the frame, cheapest candidate, arcs walked
  int changed[8];                                // per closure round (mod 8): a cost went down
  int xcc[32];                                   // XCC_ID each member saw (placement check; up to 32 members: a whole XCD)
  int n_heavy, pad1[63];                         // chunks of the frame's tokens with more than HEAVY_DEG emitting arcs (word-boundary states)
  int hist[2][4][256];                           // radix-select histograms: [max_active / min_active][round][digit]
};

struct Lay {
  Hdr* h; float* last_prob; int* mapping; int* tok_off; int* link_off; float* cost_offset;
  int* tok_state; unsigned* tok_cost; long long* tok_best; unsigned* tok_extra; unsigned* tok_prev;   // tok_best: {best link (high word), its source token}
  int* link_src; int* link_dst; int* link_arc; float* link_ac; float* link_graph; unsigned char* link_alive;
  int* gkey; int* gidx;
  Clu* clu; int* wlg; int* gkey2; int* gidx2; unsigned long long* heavy;   // heavy: 2 words per entry
};

__host__ __device__ inline size_t al(size_t v) { return (v + 255) / 256 * 256; }

__host__ __device__ __forceinline__ size_t layout(char* base, int max_frames, int max_tok, int max_link, int hash, Lay* l) {
  size_t o = 0;
  auto take = [&](size_t bytes) { char* p = base ? base + o : nullptr; o += al(bytes); return p; };
  Hdr* h = reinterpret_cast<Hdr*>(take(sizeof(Hdr)));
  float* lp = reinterpret_cast<float*>(take(sizeof(float) * MAX_C));
  int* mp = reinterpret_cast<int*>(take(sizeof(int) * (max_frames + 1)));
  int* to = reinterpret_cast<int*>(take(sizeof(int) * (max_frames + 3)));
  int* lo = reinterpret_cast<int*>(take(sizeof(int) * 2 * (max_frames + 3)));
  float* co = reinterpret_cast<float*>(take(sizeof(float) * (max_frames + 1)));
  int* ts = reinterpret_cast<int*>(take(sizeof(int) * max_tok));
  unsigned* tc = reinterpret_cast<unsigned*>(take(sizeof(unsigned) * max_tok));
  long long* tb = reinterpret_cast<long long*>(take(sizeof(long long) * max_tok));
  unsigned* te = reinterpret_cast<unsigned*>(take(sizeof(unsigned) * max_tok));
  unsigned* tp = reinterpret_cast<unsigned*>(take(sizeof(unsigned) * max_tok));
  int* ls = reinterpret_cast<int*>(take(sizeof(int) * max_link));
  int* ld = reinterpret_cast<int*>(take(sizeof(int) * max_link));
  int* la = reinterpret_cast<int*>(take(sizeof(int) * max_link));
  float* lac = reinterpret_cast<float*>(take(sizeof(float) * max_link));
  float* lg = reinterpret_cast<float*>(take(sizeof(float) * max_link));
  unsigned char* lv = reinterpret_cast<unsigned char*>(take(max_link));
  int* gk = reinterpret_cast<int*>(take(sizeof(int) * hash));
  int* gi = reinterpret_cast<int*>(take(sizeof(int) * hash));
  Clu* cl = reinterpret_cast<Clu*>(take(sizeof(Clu)));
  int* wg = reinterpret_cast<int*>(take(sizeof(int) * WLG_CAP));
  int* gk2 = reinterpret_cast<int*>(take(sizeof(int) * hash));
  int* gi2 = reinterpret_cast<int*>(take(sizeof(int) * hash));
  unsigned long long* hv = reinterpret_cast<unsigned long long*>(take(sizeof(unsigned long long) * 2 * HEAVY_CAP));
  if (l) *l = Lay{h, lp, mp, to, lo, co, ts, tc, tb, te, tp, ls, ld, la, lac, lg, lv, gk, gi, cl, wg, gk2, gi2, hv};
  return o;
}

// order-preserving float <-> unsigned (atomicMin on costs)
__device__ __forceinline__ unsigned f2o(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float o2f(unsigned o) {
  return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}

// A token's backpointer: the cheapest-arriving link with the smallest index (best_links' rule) AND that link's source token, one
// 8-byte word {link (high, signed), source token (low)} so that the best-path walk is ONE dependent load per hop; atomicMin on
// the word orders by link.  -1 = the start token, BEST_UNSET = not computed yet.
constexpr long long BEST_UNSET = 0x7fffffffffffffffLL;
__device__ __forceinline__ long long best_word(int li, int src) { return ((long long)li << 32) | (long long)(unsigned)src; }

struct Opts {
  float beam, lattice_beam, beam_delta, acoustic_scale, length_penalty, blank_skip_thresh;
  int max_active, min_active;
};

#ifdef B2T_WFST_TIMING
#define WT(i) { if (threadIdx.x == 0) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); c.tacc[i] += now_ - c.tprev; c.tprev = now_; } }
#else
#define WT(i)
#endif

constexpr int WL_CAP = 4096;   // work-list capacity (a frame that has more tokens with epsilon arcs scans all its tokens, as before)

struct Ctx {
#ifdef B2T_WFST_TIMING
  unsigned long long tacc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
#endif
  Graph g; Lay l; Opts o;
  int max_frames, max_tok, max_link, hash;
  int* key; int* idx;          // the frame's hash (LDS or HBM)
  float* ll;                   // LDS: acoustic_scale * logp of the frame
  float* redf; int* redi;      // LDS reduction scratch [NT]
  int* sh;                     // LDS scalars: [0] n_tok, [1] n_link, [2] changed, [3] overflow, [4] work-list length, [5] tokens scanned so far, [6] work list overflowed
  int* wl;                     // LDS [WL_CAP]: the frame's tokens whose state has epsilon arcs (ProcessNonemitting's work list)
};

// Block reductions: within a wave through lane permutes, across the NT / 64 waves through LDS -- two barriers instead
// of the 2 log2(NT) of a tree over the whole block (a frame makes several of them on its serial path).
__device__ __forceinline__ float block_min(Ctx& c, float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
  if ((threadIdx.x & 63) == 0) c.redf[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = c.redf[0];
#pragma unroll
  for (int w = 1; w < NT / 64; ++w) r = fminf(r, c.redf[w]);
  __syncthreads();
  return r;
}
__device__ __forceinline__ int block_sum(Ctx& c, int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  if ((threadIdx.x & 63) == 0) c.redi[threadIdx.x >> 6] = v;
  __syncthreads();
  int r = c.redi[0];
#pragma unroll
  for (int w = 1; w < NT / 64; ++w) r += c.redi[w];
  __syncthreads();
  return r;
}

// k-th smallest (0-based) of the ordered cost keys of tokens [t0, t1) (std::nth_element's value): radix select, four
// rounds of 8 bits from the top -- a 256-bin histogram in LDS per round, the bin that holds rank k found by wave 0 with a
// lane prefix sum -- instead of a 32-step bisection with a pass over the tokens and a block reduction per step.
__device__ float kth_cost(Ctx& c, int t0, int t1, int k) {
  int* hist = c.redi + 64;           // [256]; redi[0 .. 63] stay free for block_sum, redi[320 ..] hold the round's result
  int* res = c.redi + 320;           // [0] chosen digit, [1] rank inside the chosen bin
  unsigned prefix = 0u;
  int rank = k;
  for (int round = 0; round < 4; ++round) {
    const int shift = 24 - 8 * round;
    if (threadIdx.x < 256) hist[threadIdx.x] = 0;
    __syncthreads();
    for (int tb = t0; tb < t1; tb += NT) {
      const int t = tb + (int)threadIdx.x;
      const unsigned key = t < t1 ? c.l.tok_cost[t] : 0u;
      const bool act = t < t1 && (round == 0 || (key >> (shift + 8)) == (prefix >> (shift + 8)));
      const int d = (int)((key >> shift) & 255u);
      if (round < 2) {
        // the costs of a frame lie within a beam of each other: their top bits fall into a handful of bins, so the lanes
        // of a wave that share a digit send ONE LDS atomic
        unsigned long long todo = __ballot(act);
        while (todo) {
          const int leader = __ffsll((long long)todo) - 1;
          const int dl = __shfl(d, leader);
          const unsigned long long peers = __ballot(act && d == dl);
          if ((int)(threadIdx.x & 63) == leader) atomicAdd(&hist[dl], __popcll(peers));
          todo &= ~peers;
        }
      } else if (act) {
        atomicAdd(&hist[d], 1);
      }
    }
    __syncthreads();
    if (threadIdx.x < 64) {
      const int lane = threadIdx.x;
      const int h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
      const int mine = h0 + h1 + h2 + h3;
      int incl = mine;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const int up = __shfl_up(incl, o); if (lane >= o) incl += up; }
      const int excl = incl - mine;
      if (rank >= excl && rank < incl) {      // exactly one lane (0 <= rank < number of candidates)
        int r = rank - excl, d = 4 * lane;
        if (r >= h0) { r -= h0; ++d; if (r >= h1) { r -= h1; ++d; if (r >= h2) { r -= h2; ++d; } } }
        res[0] = d; res[1] = r;
      }
    }
    __syncthreads();
    prefix |= (unsigned)res[0] << shift;
    rank = res[1];
  }
  __syncthreads();
  return o2f(prefix);
}

__device__ __forceinline__ unsigned xcc_of() { unsigned x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x)); return x & 0xf; }
__device__ __forceinline__ unsigned hash_of(int state, int mask) { return ((unsigned)state * 2654435761u) & (unsigned)mask; }

// FindOrAddToken, claim phase: make sure `state` has a slot (and a token) in the frame being built; returns the slot
// (idx[slot] is valid after the next barrier) or -1 when the hash is full
__device__ __forceinline__ int claim(Ctx& c, int state) {
  const int mask = c.hash - 1;
  unsigned s = hash_of(state, mask);
  for (int probe = 0; probe < c.hash; ++probe, s = (s + 1) & mask) {
    const int k = __hip_atomic_load(&c.key[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (k == state) return (int)s;
    if (k == -1) {
      int expected = -1;
      if (__hip_atomic_compare_exchange_strong(&c.key[s], &expected, state, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
        const int id = atomicAdd(&c.sh[0], 1);
        if (id < c.max_tok) {
          c.idx[s] = id;
          c.l.tok_state[id] = state; c.l.tok_cost[id] = UMAX; c.l.tok_best[id] = BEST_UNSET; c.l.tok_extra[id] = 0u;
        } else {
          c.idx[s] = -1; atomicOr(&c.sh[3], 1);
        }
        return (int)s;
      }
      if (expected == state) return (int)s;
    }
  }
  atomicOr(&c.sh[3], 4);   // hash full
  return -1;
}
__device__ __forceinline__ int find(Ctx& c, int state) {
  const int mask = c.hash - 1;
  unsigned s = hash_of(state, mask);
  for (int probe = 0; probe < c.hash; ++probe, s = (s + 1) & mask) {
    const int k = c.key[s];
    if (k == state) return c.idx[s];
    if (k == -1) return -1;
  }
  return -1;
}

// ProcessNonemitting over the tokens [n0, ...) of the frame being built + generation of their epsilon links.
// Only the tokens of states WITH epsilon arcs matter (word ends: a few hundred of a frame's thousands), and the closure takes
// several Bellman-Ford rounds of two passes each: the tokens are classified once (token -> state -> n_eps, two dependent
// gathers) into an LDS work list that grows as the closure creates tokens; the rounds walk the list.
__device__ void nonemitting(Ctx& c, int n0, float cutoff) {
  const Graph& g = c.g;
  __syncthreads();
  if (threadIdx.x == 0) { c.sh[4] = 0; c.sh[5] = n0; c.sh[6] = 0; }
  auto extend = [&]() {                       // classify the tokens created since the last call (ends with a barrier)
    __syncthreads();
    const int from = c.sh[5], upto = min(c.sh[0], c.max_tok);
    for (int t = from + threadIdx.x; t < upto; t += NT) {
      if (g.n_eps[c.l.tok_state[t]] == 0) continue;
      const int i = atomicAdd(&c.sh[4], 1);
      if (i < WL_CAP) c.wl[i] = t; else c.sh[6] = 1;
    }
    __syncthreads();
    if (threadIdx.x == 0) c.sh[5] = upto;
    __syncthreads();
    return upto;
  };
  auto for_each = [&](int n_now, auto&& body) {   // body(t) for every token of [n0, n_now) whose state has epsilon arcs
    if (c.sh[6]) { for (int t = n0 + threadIdx.x; t < n_now; t += NT) body(t); }
    else { const int n = c.sh[4]; for (int i = threadIdx.x; i < n; i += NT) body(c.wl[i]); }
  };
  for (;;) {
    const int n_now = extend();
    if (threadIdx.x == 0) c.sh[2] = 0;
    __syncthreads();
    for (int phase = 0; phase < 2; ++phase) {
      for_each(n_now, [&](int t) {
        const int s = c.l.tok_state[t];
        const int ne = g.n_eps[s];
        if (ne == 0) return;
        const float cur = o2f(c.l.tok_cost[t]);
        if (!(cur < cutoff)) return;
        const int a0 = g.row[s];
        for (int a = a0; a < a0 + ne; ++a) {
          const float tot = cur + g_w(g, a);
          if (tot < cutoff) {
            if (phase == 0) {
              claim(c, g.next[a]);
            } else {
              const int id = find(c, g.next[a]);
              if (id >= 0) {
                const unsigned nb = f2o(tot);
                const unsigned old = atomicMin(&c.l.tok_cost[id], nb);
                if (nb < old) c.sh[2] = 1;
              }
            }
          }
        }
      });
      __syncthreads();
    }
    if (c.sh[2] == 0) break;
  }
  // forward links of the epsilon arcs, with the converged costs
  const int n_now = extend();
  for_each(n_now, [&](int t) {
    const int s = c.l.tok_state[t];
    const int ne = g.n_eps[s];
    if (ne == 0) return;
    const float cur = o2f(c.l.tok_cost[t]);
    if (!(cur < cutoff)) return;
    const int a0 = g.row[s];
    for (int a = a0; a < a0 + ne; ++a) {
      const float tot = cur + g_w(g, a);
      if (tot < cutoff) {
        const int id = find(c, g.next[a]);
        if (id < 0) continue;
        const int li = atomicAdd(&c.sh[1], 1);
        if (li < c.max_link) {
          c.l.link_src[li] = t; c.l.link_dst[li] = id; c.l.link_arc[li] = a; c.l.link_ac[li] = 0.f; c.l.link_graph[li] = g_w(g, a);
        } else {
          atomicOr(&c.sh[3], 2);
        }
      }
    }
  });
  __syncthreads();
}

// backpointers: among the links into the tokens of the new frame, the one whose cost equals the token's final cost
__device__ void best_links(Ctx& c, int l0, int l1) {
  for (int li = l0 + threadIdx.x; li < l1; li += NT) {
    const int src = c.l.link_src[li], dst = c.l.link_dst[li];
    const float tot = o2f(c.l.tok_cost[src]) + c.l.link_ac[li] + c.l.link_graph[li];
    if (f2o(tot) == c.l.tok_cost[dst]) atomicMin(&c.l.tok_best[dst], best_word(li, src));
  }
  __syncthreads();
}

__device__ void clear_hash(Ctx& c) {
  for (int i = threadIdx.x; i < c.hash; i += NT) c.key[i] = -1;
  __syncthreads();
}

// InitDecoding (:57-75)
__device__ void init_decoding(Ctx& c) {
  clear_hash(c);
  // the cluster search's scratch starts from zero whatever the caller's buffer held (its histograms are only re-zeroed AFTER a
  // frame's cut-off used them)
  for (int i = threadIdx.x; i < (int)(sizeof(Clu) / sizeof(int)); i += NT) reinterpret_cast<int*>(c.l.clu)[i] = 0;
  // ... and from two EMPTY frame hashes: the cluster search stamps its slots with the frame (below) instead of clearing a hash
  // per frame, so what an earlier utterance left in this state block must not look like a live entry of this one
  {
    unsigned long long* s0 = reinterpret_cast<unsigned long long*>(c.l.gkey);
    unsigned long long* s1 = reinterpret_cast<unsigned long long*>(c.l.gkey2);
    for (int i = threadIdx.x; i < c.hash; i += NT) { s0[i] = ~0ull; s1[i] = ~0ull; }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    Hdr* h = c.l.h;
    h->n_frames = 0; h->overflow = 0; h->num_input = 0; h->is_last_blank = 0; h->last_best = 0; h->finalized = 0;
    h->final_best = 0.f; h->has_final = 0; h->arcs_lo = 0u; h->arcs_hi = 0u;
    h->links_marked = 0; h->n_prunes = 0; h->peak_tok = 0; h->peak_link = 0; h->removed_tok = 0; h->removed_link = 0;
    c.l.clu->bar = 0u; c.l.clu->bar_base = 0u; c.l.clu->overflow = 0;
    c.sh[0] = 0; c.sh[1] = 0; c.sh[3] = 0;
    c.l.tok_off[0] = 0;
    c.l.link_off[0] = 0;
  }
  __syncthreads();
  if (threadIdx.x == 0) claim(c, c.g.start);
  __syncthreads();
  if (threadIdx.x == 0) { c.l.tok_cost[0] = f2o(0.f); c.l.tok_best[0] = -1; }
  __syncthreads();
  nonemitting(c, 0, c.o.beam);
  best_links(c, 0, min(c.sh[1], c.max_link));
  if (threadIdx.x == 0) {
    c.l.tok_best[0] = -1;
    c.l.tok_off[1] = min(c.sh[0], c.max_tok);
    c.l.link_off[1] = min(c.sh[1], c.max_link);    // [eps links of frame 0]
    c.l.h->n_tok = c.sh[0]; c.l.h->n_link = c.sh[1]; c.l.h->overflow = c.sh[3];
  }
  __syncthreads();
}

// one AdvanceDecoding(.., 1): ProcessEmitting + ProcessNonemitting on the row `logp` (already in c.ll, scaled)
__device__ void advance(Ctx& c) {
  const Graph& g = c.g;
  const int f = c.l.h->n_frames;
  if (f >= c.max_frames) { if (threadIdx.x == 0) atomicOr(&c.sh[3], 8); __syncthreads(); return; }
  const int t0 = c.l.tok_off[f], t1 = c.l.tok_off[f + 1];
  WT(0)
  // ---- GetCutoff (:650-720)
  float best = INFINITY;
  for (int t = t0 + threadIdx.x; t < t1; t += NT) best = fminf(best, o2f(c.l.tok_cost[t]));
  best = block_min(c, best);
  const int n = t1 - t0;
  const float beam_cutoff = best + c.o.beam;
  float cur_cutoff = beam_cutoff, adaptive = c.o.beam;
  {
    float max_cut = INFINITY, min_cut = INFINITY;
    if (n > c.o.max_active) max_cut = kth_cost(c, t0, t1, c.o.max_active);
    if (max_cut < beam_cutoff) {
      cur_cutoff = max_cut; adaptive = max_cut - best + c.o.beam_delta;
    } else {
      if (n > c.o.min_active) min_cut = c.o.min_active == 0 ? best : kth_cost(c, t0, t1, c.o.min_active);
      if (min_cut > beam_cutoff) { cur_cutoff = min_cut; adaptive = min_cut - best + c.o.beam_delta; }
    }
  }
  WT(1)   // best + k-th cost
  const float cost_offset = -best;
  const float lp = c.o.length_penalty;
  // ---- ProcessEmitting (:722-824).  Out-degrees are skewed (1-3 arcs inside a word, hundreds at the word-boundary states
  // of L o G, each arc a dependent chain of gathers + a hash probe), so a thread that owned a token would idle most of its
  // wave.  A wave takes 64 consecutive tokens, scans their degrees through lane permutes and walks the FLATTENED arc list 64
  // arcs at a time: arc j belongs to the first lane whose inclusive degree sum exceeds j (6-step search through permutes).
  float mn = INFINITY;
  int narcs = 0;
  auto arc_cost = [&](float cur, int s, int a, float& ac, float& gc) {
    ac = cost_offset - c.ll[g_il(g, a) - 1];
    gc = g_w(g, a);
    if (lp != 0.f && g.next[a] != s) gc += lp;     // (no gather of the destination when there is no length penalty)
    return cur + ac + gc;
  };
  const int lane = threadIdx.x & 63;
  auto walk = [&](auto&& visit) {          // visit(token, its cost, its state, arc) for every emitting arc of every token under the cutoff
    for (int base = t0 + (int)threadIdx.x - lane; base < t1; base += NT) {    // wave-uniform
      const int t = base + lane;
      float cur = INFINITY; int s = 0, a0 = 0, deg = 0;
      if (t < t1) {
        cur = o2f(c.l.tok_cost[t]);
        if (cur <= cur_cutoff) { s = c.l.tok_state[t]; a0 = g.row[s] + g.n_eps[s]; deg = g.row[s + 1] - a0; }
      }
      int incl = deg;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(incl, off, 64); if (lane >= off) incl += v; }
      const int total = __shfl(incl, 63, 64), excl = incl - deg;
      for (int jb = 0; jb < total; jb += 64) {
        const int jj = jb + lane;
        int owner = 0;                      // number of lanes whose inclusive sum is <= jj
#pragma unroll
        for (int step = 32; step >= 1; step >>= 1) { const int v = __shfl(incl, owner + step - 1, 64); if (v <= jj) owner += step; }
        owner = min(owner, 63);
        const int oa0 = __shfl(a0, owner, 64), oex = __shfl(excl, owner, 64), os = __shfl(s, owner, 64);
        const float ocur = __shfl(cur, owner, 64);
        if (jj < total) visit(base + owner, ocur, os, oa0 + (jj - oex));
      }
      narcs += deg;
    }
  };
  // Pass A: the frame's best candidate -> next_cutoff
  walk([&](int, float cur, int s, int a) { float ac, gc; mn = fminf(mn, arc_cost(cur, s, a, ac, gc)); });
  mn = block_min(c, mn);
  narcs = block_sum(c, narcs);
  if (threadIdx.x == 0) {
    const unsigned lo = c.l.h->arcs_lo + (unsigned)narcs;
    if (lo < c.l.h->arcs_lo) c.l.h->arcs_hi += 1u;
    c.l.h->arcs_lo = lo;
  }
  const float next_cutoff = mn + adaptive;
  WT(2)   // pass A
  clear_hash(c);
  WT(3)   // clear hash
  const int n0 = min(c.sh[0], c.max_tok), l0 = min(c.sh[1], c.max_link);
  // pass B.  Phase 0 walks the arcs once: every surviving arc claims its destination's hash slot and is recorded as a
  // forward link that still names the SLOT (token ids are handed out by the claim's winner and are only safe to read after
  // a barrier).  Phase 1 is a flat, perfectly balanced loop over those links: slot -> token id, cost minimisation.
  {
    auto visit = [&](int t, float cur, int s, int a) {
      float ac, gc;
      const float tot = arc_cost(cur, s, a, ac, gc);
      if (!(tot < next_cutoff)) return;
      const int slot = claim(c, g.next[a]);
      if (slot < 0) return;
      const int li = atomicAdd(&c.sh[1], 1);
      if (li < c.max_link) {
        c.l.link_src[li] = t; c.l.link_dst[li] = slot; c.l.link_arc[li] = a; c.l.link_ac[li] = ac; c.l.link_graph[li] = gc;
      } else {
        atomicOr(&c.sh[3], 2);
      }
    };
    { const int keep = narcs; walk(visit); narcs = keep; }
    __syncthreads();
    WT(4)   // pass B: arc walk (claim + link records)
    const int l1 = min(c.sh[1], c.max_link);
    for (int li = l0 + threadIdx.x; li < l1; li += NT) {
      int id = c.idx[c.l.link_dst[li]];
      if (id < 0) id = 0;                  // token capacity exceeded: the overflow bit is set and the caller discards the result
      const float tot = o2f(c.l.tok_cost[c.l.link_src[li]]) + c.l.link_ac[li] + c.l.link_graph[li];
      c.l.link_dst[li] = id;
      atomicMin(&c.l.tok_cost[id], f2o(tot));
    }
    __syncthreads();
    WT(5)   // pass B: link walk (token ids, costs)
  }
  if (threadIdx.x == 0) c.l.link_off[2 * f + 2] = min(c.sh[1], c.max_link);   // [emitting links f -> f+1]
  __syncthreads();
  nonemitting(c, n0, next_cutoff);
  WT(6)   // epsilon closure + links
  best_links(c, l0, min(c.sh[1], c.max_link));
  WT(7)   // best links
  if (threadIdx.x == 0) {
    c.l.cost_offset[f] = cost_offset;
    c.l.tok_off[f + 2] = min(c.sh[0], c.max_tok);
    c.l.link_off[2 * f + 3] = min(c.sh[1], c.max_link);                       // [eps links of frame f+1]
    c.l.h->n_frames = f + 1;
    c.l.h->n_tok = c.sh[0]; c.l.h->n_link = c.sh[1]; c.l.h->overflow = c.sh[3];
  }
  __syncthreads();
}

__device__ void setup(Ctx& c, const Graph& g, char* state, int u, size_t state_bytes, const Opts& o, int max_frames, int max_tok,
                      int max_link, int hash, int* smem_hash, float* ll, float* redf, int* redi, int* sh, int* wl) {
  c.g = g; c.o = o; c.max_frames = max_frames; c.max_tok = max_tok; c.max_link = max_link; c.hash = hash;
  layout(state + (size_t)u * state_bytes, max_frames, max_tok, max_link, hash, &c.l);
  if (smem_hash) { c.key = smem_hash; c.idx = smem_hash + hash; } else { c.key = c.l.gkey; c.idx = c.l.gidx; }
  c.ll = ll; c.redf = redf; c.redi = redi; c.sh = sh; c.wl = wl;
}


// =====================================================================================================================
// CLUSTER search: G workgroups (G = 2, 4 or 8) per utterance instead of one, so that 32 utterances use the whole chip
// instead of 32 of its 256 CUs.  The G workgroups of an utterance are placed on ONE XCD (block b runs on XCD b % 8; checked
// at run time through XCC_ID), i.e. behind one L2:
//   * the frame's token hash, the tokens, the links and a handful of counters live in the utterance's state block and are
//     shared through that L2: plain stores (write-through the CU's vector cache into L2), L2 atomics (hash CAS, cost
//     atomicMin, counters) and L1-bypassing sc1 loads for everything another workgroup may have written;
//   * a frame is a sequence of phases separated by CLUSTER barriers (a monotonic arrival counter in L2, one lane per
//     workgroup arrives and polls) -- 6 per frame, + 4 when max_active binds -- instead of the ~45 workgroup barriers of the
//     single-workgroup kernel: claim and relax are ONE phase (the claim's winner publishes the token id AFTER the token's
//     fields have reached L2; a loser polls the slot), the epsilon work list is appended to by whoever creates a token, the
//     frame's best cost is kept by atomicMin while costs are written, two hashes alternate so that clearing one hides
//     under pass A, and the backpointer pass of a frame runs inside pass A of the next.
// Same arithmetic and the same results as wfst_search_kernel (tests/test_gpu_wfst.py runs both against the oracle).
// =====================================================================================================================
constexpr int UNSET = -2;
constexpr unsigned CBAR_SPIN_LIMIT = 1u << 22;

__device__ __forceinline__ int ldi(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned ldu(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ldf(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

#ifdef B2T_WFST_TIMING
#define CT(i) { if (c.gtid == 0) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); c.tacc[i] += now_ - c.tprev; c.tprev = now_; } }
#else
#define CT(i)
#endif

struct CCtx {
#ifdef B2T_WFST_TIMING
  unsigned long long tacc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
#endif
  Graph g; Lay l; Opts o; Clu* cl;
  int max_frames, max_tok, max_link, hash;
  int G, j, gtid, gthreads;
  unsigned bar_target;
  float* ll; float* redf; int* redi; int* lsh;   // LDS: frame log-likelihoods, reduction scratch, [0] dead flag, [1..] scalars
  int* key; int* idx;                            // hash of the frame being built
  // Frame-stamped slots (round 4): the key half of a slot is stamp << 27 | state, a slot whose stamp is not the current frame's
  // reads as empty -- no hash is cleared per frame any more (0.42 GB of the 1.13 GB a 25-frame launch of 32 utterances wrote).
  // Stamps 1 .. 30 cycle over a hash's uses (0 = zeroed memory, 31 = cleared marker: never live), so a hash is cleared once per
  // 30 uses.  Needs states < 2^27; larger graphs (stamped = 0) clear per frame as before.
  int stamped; unsigned stamp;
  int* stk_t; float* stk_c;                      // LDS: per-thread stack of the epsilon closure's chase ([CHASE_DEPTH][NT])
};

// Cluster barrier.  Every store this workgroup issued has reached L2 (vmcnt(0): stores are acknowledged by L2) before its
// arrival is counted; readers use sc1 loads, so nothing has to be invalidated.  Returns false after a timeout (a member is
// not resident or died): the overflow word gets bit 32 and every member leaves at its next barrier.
__device__ bool cbar(CCtx& c) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  c.bar_target += (unsigned)c.G;
  if (threadIdx.x == 0 && !c.lsh[0]) {
    __hip_atomic_fetch_add(&c.cl->bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while ((int)(ldu(&c.cl->bar) - c.bar_target) < 0) {
      if (++spins > CBAR_SPIN_LIMIT || ((spins & 1023u) == 0u && (ldi(&c.cl->overflow) & 32))) { atomicOr(&c.cl->overflow, 32); c.lsh[0] = 1; break; }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  __syncthreads();
  return c.lsh[0] == 0;
}

__device__ __forceinline__ float cblock_min(CCtx& c, float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
  if ((threadIdx.x & 63) == 0) c.redf[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = c.redf[0];
#pragma unroll
  for (int w = 1; w < NT / 64; ++w) r = fminf(r, c.redf[w]);
  __syncthreads();
  return r;
}
__device__ __forceinline__ int cblock_sum(CCtx& c, int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  if ((threadIdx.x & 63) == 0) c.redi[threadIdx.x >> 6] = v;
  __syncthreads();
  int r = c.redi[0];
#pragma unroll
  for (int w = 1; w < NT / 64; ++w) r += c.redi[w];
  __syncthreads();
  return r;
}

// k-th smallest cost of the tokens [t0, t1): the radix select of kth_cost with the histogram of a round summed over the
// cluster in L2 (every member then picks the digit from the same 256 numbers): one cluster barrier per round.
__device__ float ckth_cost(CCtx& c, int t0, int t1, int k, int set, bool& ok) {
  int* hist = c.redi + 64;
  int* res = c.redi + 320;
  unsigned prefix = 0u;
  int rank = k;
  for (int round = 0; round < 4; ++round) {
    const int shift = 24 - 8 * round;
    if (threadIdx.x < 256) hist[threadIdx.x] = 0;
    __syncthreads();
    for (int tb = t0 + c.j * NT; tb < t1; tb += c.gthreads) {
      const int t = tb + (int)threadIdx.x;
      const unsigned key = t < t1 ? ldu(&c.l.tok_cost[t]) : 0u;
      const bool act = t < t1 && (round == 0 || (key >> (shift + 8)) == (prefix >> (shift + 8)));
      const int d = (int)((key >> shift) & 255u);
      if (round < 2) {
        unsigned long long todo = __ballot(act);
        while (todo) {
          const int leader = __ffsll((long long)todo) - 1;
          const int dl = __shfl(d, leader);
          const unsigned long long peers = __ballot(act && d == dl);
          if ((int)(threadIdx.x & 63) == leader) atomicAdd(&hist[dl], __popcll(peers));
          todo &= ~peers;
        }
      } else if (act) {
        atomicAdd(&hist[d], 1);
      }
    }
    __syncthreads();
    if (threadIdx.x < 256 && hist[threadIdx.x]) atomicAdd(&c.cl->hist[set][round][threadIdx.x], hist[threadIdx.x]);
    if (!cbar(c)) { ok = false; return 0.f; }
    if (threadIdx.x < 64) {
      const int lane = threadIdx.x;
      const int* gh = c.cl->hist[set][round];
      const int h0 = ldi(gh + 4 * lane), h1 = ldi(gh + 4 * lane + 1), h2 = ldi(gh + 4 * lane + 2), h3 = ldi(gh + 4 * lane + 3);
      const int mine = h0 + h1 + h2 + h3;
      int incl = mine;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const int up = __shfl_up(incl, o); if (lane >= o) incl += up; }
      const int excl = incl - mine;
      if (rank >= excl && rank < incl) {
        int r = rank - excl, d = 4 * lane;
        if (r >= h0) { r -= h0; ++d; if (r >= h1) { r -= h1; ++d; if (r >= h2) { r -= h2; ++d; } } }
        res[0] = d; res[1] = r;
      }
    }
    __syncthreads();
    prefix |= (unsigned)res[0] << shift;
    rank = res[1];
    __syncthreads();
  }
  return o2f(prefix);
}

// One slot per ACTIVE lane from a shared counter with ONE atomic per wave: a counter word in L2 serves ~90 atomics per
// microsecond, and a frame allocates ~50 k links and ~8 k tokens (one atomic each: 60 ms of the first version's 72).
__device__ __forceinline__ int wave_alloc(int* counter) {
  const unsigned long long m = __ballot(1);
  const int lane = threadIdx.x & 63;
  const int leader = __ffsll((long long)m) - 1;
  int base = 0;
  if (lane == leader) base = atomicAdd(counter, __popcll(m));
  base = __shfl(base, leader, 64);
  return base + __popcll(m & ((1ull << lane) - 1ull));
}

// FindOrAddToken across the cluster: returns the token id of `state` in the frame being built (-1: hash or token capacity
// exhausted).  A hash slot is ONE 8-byte word {state, token id} (the two int arrays of the layout are contiguous), so the
// common case -- the token exists -- is a single L1-bypassing 8-byte load.  The CAS (on the state half) winner allocates the
// token, writes its fields, waits until they are in L2 and only then publishes the id in the other half; everyone else polls
// the word.  A new token whose state has epsilon arcs joins the work list.
template <bool ST>   // ST: frame-stamped slots (compile-time: both claim paths in one kernel spilled 536 B per lane to scratch, 13.6 -> 21.7 ms)
__device__ __forceinline__ int cclaim(CCtx& c, int state) {
  const int mask = c.hash - 1;
  unsigned s = hash_of(state, mask);
  unsigned long long* slots = reinterpret_cast<unsigned long long*>(c.key);
  if constexpr (ST) {
    const unsigned want = (c.stamp << 27) | (unsigned)state;
    for (int probe = 0; probe < c.hash; ++probe, s = (s + 1) & mask) {
      unsigned long long v = __hip_atomic_load(&slots[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (;;) {
        const unsigned k = (unsigned)(v & 0xffffffffull);
        if ((k >> 27) != c.stamp) {               // not of this frame: empty.  ONE 64-bit CAS takes the slot and unsets the stale id with it
          const unsigned long long mine = ((unsigned long long)(unsigned)UNSET << 32) | want;
          const unsigned long long seen = atomicCAS(&slots[s], v, mine);   // (the value-returning form: taking &v for the builtin's `expected` put the loop's state into scratch memory, 13.6 -> 21.7 ms)
          const bool won = seen == v;
          v = seen;
          if (won) {
            int id = wave_alloc(&c.cl->n_tok);
            if (id < c.max_tok) {
              c.l.tok_state[id] = state; c.l.tok_cost[id] = UMAX; c.l.tok_best[id] = BEST_UNSET; c.l.tok_extra[id] = 0u;
              if (c.g.n_eps[state] > 0) {
                const int w = wave_alloc(&c.cl->wl_n);
                if (w < WLG_CAP) c.l.wlg[w] = id; else atomicOr(&c.cl->overflow, 16);
              }
              asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
              atomicOr(&c.cl->overflow, 1); id = -1;
            }
            __hip_atomic_store(reinterpret_cast<int*>(&slots[s]) + 1, id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return id;
          }
          continue;                                 // lost the race: v holds what is there now, look at it again
        }
        if (k == want) {
          int id = (int)(unsigned)(v >> 32), spins = 0;
          while (id == UNSET) {
            if (++spins > (1 << 24)) { atomicOr(&c.cl->overflow, 32); return -1; }
            id = (int)(unsigned)(__hip_atomic_load(&slots[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 32);
          }
          return id;
        }
        break;                                      // another state of this frame: next slot
      }
    }
    atomicOr(&c.cl->overflow, 4);
    return -1;
  }
  for (int probe = 0; probe < c.hash; ++probe, s = (s + 1) & mask) {
    unsigned long long v = __hip_atomic_load(&slots[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int k = (int)(unsigned)(v & 0xffffffffull);
    if (k == -1) {
      int* kp = reinterpret_cast<int*>(&slots[s]);
      int expected = -1;
      if (__hip_atomic_compare_exchange_strong(kp, &expected, state, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
        int id = wave_alloc(&c.cl->n_tok);
        if (id < c.max_tok) {
          c.l.tok_state[id] = state; c.l.tok_cost[id] = UMAX; c.l.tok_best[id] = BEST_UNSET; c.l.tok_extra[id] = 0u;
          if (c.g.n_eps[state] > 0) {
            const int w = wave_alloc(&c.cl->wl_n);
            if (w < WLG_CAP) c.l.wlg[w] = id; else atomicOr(&c.cl->overflow, 16);
          }
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
          atomicOr(&c.cl->overflow, 1); id = -1;
        }
        __hip_atomic_store(kp + 1, id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return id;
      }
      k = expected;
      v = ((unsigned long long)(unsigned)UNSET << 32) | (unsigned)k;
    }
    if (k == state) {
      int id = (int)(unsigned)(v >> 32), spins = 0;
      while (id == UNSET) {
        if (++spins > (1 << 24)) { atomicOr(&c.cl->overflow, 32); return -1; }
        id = (int)(unsigned)(__hip_atomic_load(&slots[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 32);
      }
      return id;
    }
  }
  atomicOr(&c.cl->overflow, 4);
  return -1;
}

// best_links over the links [l0, l1) (deferred: runs in pass A of the next frame / at the end of the launch)
__device__ __forceinline__ void cbest_links(CCtx& c, int l0, int l1) {
  for (int li = l0 + c.gtid; li < l1; li += c.gthreads) {
    const int src = ldi(&c.l.link_src[li]), dst = ldi(&c.l.link_dst[li]);
    const float tot = o2f(ldu(&c.l.tok_cost[src])) + ldf(&c.l.link_ac[li]) + ldf(&c.l.link_graph[li]);
    if (f2o(tot) == ldu(&c.l.tok_cost[dst])) atomicMin(&c.l.tok_best[dst], best_word(li, src));
  }
}

struct CFrame { int f, t0, t1, pl0, pl1; };   // decoded frames so far, tokens of the newest frame, links awaiting best_links

// One AdvanceDecoding(.., 1) by the whole cluster.  fr is cluster-uniform private state, updated on success.
template <bool ST, bool CP>
__device__ bool cadvance(CCtx& c, CFrame& fr) {
  const Graph& g = c.g;
  Clu* cl = c.cl;
  const int f = fr.f, t0 = fr.t0, t1 = fr.t1, par = f & 1, npar = par ^ 1;
  if (f >= c.max_frames) { if (c.gtid == 0) atomicOr(&cl->overflow, 8); return cbar(c) && false; }
  CT(0)
  // ---- GetCutoff (:650-720): the frame's best cost was kept by atomicMin while the costs were written
  const float best = o2f(ldu(&cl->best[par]));
  const int n = t1 - t0;
  const float beam_cutoff = best + c.o.beam;
  float cur_cutoff = beam_cutoff, adaptive = c.o.beam;
  {
    bool ok = true;
    float max_cut = INFINITY, min_cut = INFINITY;
    if (n > c.o.max_active) { max_cut = ckth_cost(c, t0, t1, c.o.max_active, 0, ok); if (!ok) return false; }
    if (max_cut < beam_cutoff) {
      cur_cutoff = max_cut; adaptive = max_cut - best + c.o.beam_delta;
    } else {
      if (n > c.o.min_active) {
        if (c.o.min_active == 0) min_cut = best;
        else { min_cut = ckth_cost(c, t0, t1, c.o.min_active, 1, ok); if (!ok) return false; }
      }
      if (min_cut > beam_cutoff) { cur_cutoff = min_cut; adaptive = min_cut - best + c.o.beam_delta; }
    }
  }
  CT(1)   // cutoff (k-th cost)
  const float cost_offset = -best;
  const float lp = c.o.length_penalty;
  auto arc_cost = [&](float cur, int s, int a, float& ac, float& gc) {
    ac = cost_offset - c.ll[g_il_t<CP>(g, a) - 1];
    gc = g_w_t<CP>(g, a);
    if (lp != 0.f && g.next[a] != s) gc += lp;
    return cur + ac + gc;
  };
  const int lane = threadIdx.x & 63;
  int narcs = 0;
  // Work distribution.  Out-degrees are bimodal: ~3 arcs inside a word, hundreds at the word-boundary states of L o G, and
  // the word-boundary tokens sit together at the end of a frame's token range (the epsilon closure creates them last).
  //   light tokens (<= HEAVY_DEG arcs): 64-token blocks dealt round-robin over ALL waves of the cluster (block q -> member
  //     q % G), arcs flattened inside the wave as in the single-workgroup kernel;
  //   heavy tokens: pass A's light walk lists them CHUNK by chunk (64 arcs, up to 16 chunks; longer rows get wider chunks),
  //     then one wave per chunk, round-robin: a 400-arc token is seven waves' work, not seven trips of one wave while its
  //     neighbours idle (~66 heavy tokens per frame for 128 waves).
  // (With the blocks dealt member by member the members that got the frame's last blocks took 3-4x as long as the others.)
  const int gwave = (int)(threadIdx.x >> 6) * c.G + c.j, nwaves = c.G * (NT / 64);
  auto walk_light = [&](bool collect, auto&& visit) {
    for (int base = t0 + gwave * 64; base < t1; base += nwaves * 64) {    // wave-uniform
      const int t = base + lane;
      float cur = INFINITY; int s = 0, a0 = 0, deg = 0;
      if (t < t1) {
        cur = o2f(ldu(&c.l.tok_cost[t]));
        if (cur <= cur_cutoff) { s = ldi(&c.l.tok_state[t]); a0 = g.row[s] + g.n_eps[s]; deg = g.row[s + 1] - a0; }
      }
      narcs += deg;
      int nch = 0, sh = 0;
      if (deg > HEAVY_DEG) {
        if (collect) {                             // chunks of 64 << sh arcs, at most 16 per token
          while (((deg - 1) >> (6 + sh)) >= 16) ++sh;
          nch = ((deg - 1) >> (6 + sh)) + 1;
        }
        deg = 0;
      }
      if (collect && __ballot(nch > 0)) {          // wave-uniform: one counter atomic per wave for all its chunks
        int ci = nch;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(ci, off, 64); if (lane >= off) ci += v; }
        const int ctot = __shfl(ci, 63, 64);
        int cb = 0;
        if (lane == 0) cb = atomicAdd(&cl->n_heavy, ctot);
        cb = __shfl(cb, 0, 64) + ci - nch;
        // (an entry carries everything a walk needs -- the token's cost is final by now --: the walks below go from the entry
        //  straight to the arcs, two dependent round trips fewer than through tok_cost / tok_state / row)
        const int hspan = 64 << sh, ha0 = a0, hdeg = g.row[s + 1] - a0;
        for (int k = 0; k < nch; ++k) {
          if (cb + k < HEAVY_CAP) {
            const int ab = ha0 + k * hspan, ae = ha0 + min(hdeg, (k + 1) * hspan);
            c.l.heavy[2 * (cb + k)] = ((unsigned long long)__float_as_uint(cur) << 32) | (unsigned)t;
            c.l.heavy[2 * (cb + k) + 1] = ((unsigned long long)(unsigned)ae << 32) | (unsigned)ab;
          } else {
            atomicOr(&cl->overflow, 16);           // (128 k chunks in one frame: capacity error; every entry below the cap is written)
          }
        }
      }
      int incl = deg;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(incl, off, 64); if (lane >= off) incl += v; }
      const int total = __shfl(incl, 63, 64), excl = incl - deg;
      for (int jb = 0; jb < total; jb += 64) {
        const int jj = jb + lane;
        int owner = 0;
#pragma unroll
        for (int step = 32; step >= 1; step >>= 1) { const int v = __shfl(incl, owner + step - 1, 64); if (v <= jj) owner += step; }
        owner = min(owner, 63);
        const int oa0 = __shfl(a0, owner, 64), oex = __shfl(excl, owner, 64), os = __shfl(s, owner, 64);
        const float ocur = __shfl(cur, owner, 64);
        if (jj < total) visit(base + owner, ocur, os, oa0 + (jj - oex));
      }
    }
  };
  auto walk_heavy = [&](auto&& visit) {
    const int nh = min(ldi(&cl->n_heavy), HEAVY_CAP);
#ifdef B2T_WFST_TIMING
    if (c.gtid == 0) { c.tacc[12] += nh; c.tacc[13] += 1; }
#endif
    for (int i = gwave; i < nh; i += nwaves) {
      const unsigned long long e0 = __hip_atomic_load(&c.l.heavy[2 * i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned long long e1 = __hip_atomic_load(&c.l.heavy[2 * i + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int t = (int)(unsigned)(e0 & 0xffffffffull), ab = (int)(unsigned)(e1 & 0xffffffffull), ae = (int)(unsigned)(e1 >> 32);
      const float cur = __uint_as_float((unsigned)(e0 >> 32));
      const int s = lp != 0.f ? ldi(&c.l.tok_state[t]) : -1;      // (only the length penalty looks at the source state)
      for (int jb = ab; jb < ae; jb += 64) if (jb + lane < ae) visit(t, cur, s, jb + lane);
    }
  };
  // ---- pass A: the frame's cheapest candidate.  Under it: the previous frame's backpointers, the other hash cleared,
  //      the next parity's accumulators and the histograms reset.
  float mn = INFINITY;
  auto visit_a = [&](int, float cur, int s, int a) { float ac, gc; mn = fminf(mn, arc_cost(cur, s, a, ac, gc)); };
  walk_light(true, visit_a);
  CT(11)  // pass A, light tokens
  if (!cbar(c)) return false;                    // the heavy list is complete
  walk_heavy(visit_a);
  mn = cblock_min(c, mn);
  narcs = cblock_sum(c, narcs);
  if (threadIdx.x == 0) { atomicMin(&cl->cand_min[par], f2o(mn)); atomicAdd(&cl->narcs[par], narcs); }
  CT(2)   // pass A walk
  cbest_links(c, fr.pl0, fr.pl1);
  CT(3)   // deferred best links
  int* nkey = npar ? c.l.gkey2 : c.l.gkey; int* nidx = nkey + c.hash;    // (key / idx arrays are adjacent: 8-byte slots)
  const unsigned use = (unsigned)(f + 1) >> 1;     // how often this hash has been used before (frame f + 1 is built in hash (f + 1) & 1)
  c.stamp = 1u + use % 30u;
  if (!ST || (c.stamp == 1u && use > 0u)) {   // stamped: only when the stamps wrap; InitDecoding left both hashes empty
    unsigned long long* ns = reinterpret_cast<unsigned long long*>(nkey);
    const unsigned long long empty = ((unsigned long long)(unsigned)UNSET << 32) | 0xffffffffull;
    for (int i = c.gtid; i < c.hash; i += c.gthreads) ns[i] = empty;
  }
  if (c.j == 0) {
    int* hz = &cl->hist[0][0][0];
    for (int i = threadIdx.x; i < 2 * 4 * 256; i += NT) hz[i] = 0;
    if (threadIdx.x < 8) cl->changed[threadIdx.x] = 0;
    if (threadIdx.x == 0) { cl->best[npar] = UMAX; cl->wl_n = 0; }
  }
  CT(4)   // clears
  if (!cbar(c)) return false;
  CT(5)   // barrier A
  c.key = nkey; c.idx = nidx;
  const unsigned cmin = ldu(&cl->cand_min[par]);
  const float next_cutoff = (cmin == UMAX ? INFINITY : o2f(cmin)) + adaptive;
  const int n0 = ldi(&cl->n_tok), l0 = ldi(&cl->n_link);
  if (c.gtid == 0) {
    const unsigned na = (unsigned)ldi(&cl->narcs[par]);
    const unsigned lo = c.l.h->arcs_lo + na;
    if (lo < c.l.h->arcs_lo) c.l.h->arcs_hi += 1u;
    c.l.h->arcs_lo = lo;
  }
  // ---- pass B: claim + relax + link record in one walk
  auto visit_b = [&](int t, float cur, int s, int a) {
    float ac, gc;
    const float tot = arc_cost(cur, s, a, ac, gc);
    if (!(tot < next_cutoff)) return;
    const int id = cclaim<ST>(c, g.next[a]);
    if (id < 0) return;
    const unsigned nb = f2o(tot);
    atomicMin(&c.l.tok_cost[id], nb);
    const int li = wave_alloc(&cl->n_link);
    if (li < c.max_link) {
      c.l.link_src[li] = t; c.l.link_dst[li] = id; c.l.link_arc[li] = a; c.l.link_ac[li] = ac; c.l.link_graph[li] = gc;
    } else {
      atomicOr(&cl->overflow, 2);
    }
  };
  walk_light(false, visit_b);
  walk_heavy(visit_b);
  if (c.gtid == 0) { cl->cand_min[npar] = UMAX; cl->narcs[npar] = 0; }   // (read above by everyone, not needed before frame f + 1's pass A)
  CT(6)   // pass B walk
  if (!cbar(c)) return false;
  CT(7)   // barrier B
  const int lem = min(ldi(&cl->n_link), c.max_link);
  // ---- ProcessNonemitting.  One pass over the work list in which whoever LOWERS a token's cost goes on to relax that token's
  //      epsilon arcs itself, with the value it wrote (a small per-thread stack in LDS): every final cost was written by a
  //      thread that then relaxed the token's arcs with exactly that cost, so the pass ends at the fixed point and needs no
  //      second sweep to notice it -- one cluster barrier instead of one per level of the epsilon chains plus one (4-5 rounds
  //      of ~10 us each before).  Only a stack overflow (CHASE_DEPTH pending tokens in one thread) asks for another round.
  for (int round = 0;; ++round) {
    const int wn = min(ldi(&cl->wl_n), WLG_CAP);
    if (round > 0 && ldi(&cl->changed[(round - 1) & 7]) == 0) break;
    if (c.gtid == 0) cl->changed[(round + 2) & 7] = 0;
    for (int i = c.gtid; i < wn; i += c.gthreads) {
      // (the stack holds STATES and the costs written for them: relaxing a token's arcs needs nothing else, so a chased token
      //  costs no load of its own -- a level of the chain is row -> arc -> {hash slot, n_eps of the target} -> cost atomic)
      int sp = 1, pops = 0;
      {
        const int t = ldi(&c.l.wlg[i]);
        const unsigned c0 = ldu(&c.l.tok_cost[t]);
        c.stk_t[threadIdx.x] = ldi(&c.l.tok_state[t]);
        c.stk_c[threadIdx.x] = o2f(c0);
      }
      while (sp > 0) {
        if (++pops > (1 << 14)) { atomicOr(&cl->overflow, 32); break; }   // (an epsilon cycle of negative weight: refuse, do not hang)
        --sp;
        const int s = c.stk_t[sp * NT + threadIdx.x];
        const float cur = c.stk_c[sp * NT + threadIdx.x];
        if (!(cur < next_cutoff)) continue;
        const int a0 = g.row[s], ne = g.n_eps[s];
        for (int a = a0; a < a0 + ne; ++a) {
          const float tot = cur + g_w_t<CP>(g, a);
          if (tot < next_cutoff) {
            const int ns = g.next[a];
            const int nne = g.n_eps[ns];           // (in flight next to the claim's slot load)
            const int id = cclaim<ST>(c, ns);
            if (id < 0) continue;
            const unsigned nb = f2o(tot);
            const unsigned old = atomicMin(&c.l.tok_cost[id], nb);
            if (nb < old && nne > 0) {
              if (sp < CHASE_DEPTH) { c.stk_t[sp * NT + threadIdx.x] = ns; c.stk_c[sp * NT + threadIdx.x] = tot; ++sp; }
              else __hip_atomic_store(&cl->changed[round & 7], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
          }
        }
      }
    }
    if (!cbar(c)) return false;
    if (round > 4096) { atomicOr(&cl->overflow, 32); break; }
  }
  CT(8)   // closure rounds incl. their barriers
  // ---- the epsilon links with the converged costs, and the new frame's best cost
  {
    const int wn = min(ldi(&cl->wl_n), WLG_CAP);
    for (int i = c.gtid; i < wn; i += c.gthreads) {
      const int t = ldi(&c.l.wlg[i]);
      const int s = ldi(&c.l.tok_state[t]);
      const float cur = o2f(ldu(&c.l.tok_cost[t]));
      if (!(cur < next_cutoff)) continue;
      const int a0 = g.row[s], ne = g.n_eps[s];
      for (int a = a0; a < a0 + ne; ++a) {
        const float tot = cur + g_w_t<CP>(g, a);
        if (tot < next_cutoff) {
          const int id = cclaim<ST>(c, g.next[a]);      // exists: the closure has converged
          if (id < 0) continue;
          const int li = wave_alloc(&cl->n_link);
          if (li < c.max_link) {
            c.l.link_src[li] = t; c.l.link_dst[li] = id; c.l.link_arc[li] = a; c.l.link_ac[li] = 0.f; c.l.link_graph[li] = g_w_t<CP>(g, a);
          } else {
            atomicOr(&cl->overflow, 2);
          }
        }
      }
    }
    if (c.gtid == 0) cl->n_heavy = 0;            // (last read in pass B; next written in the next frame's pass A)
    float b2 = INFINITY;
    const int n1 = min(ldi(&cl->n_tok), c.max_tok);
    for (int t = n0 + c.gtid; t < n1; t += c.gthreads) b2 = fminf(b2, o2f(ldu(&c.l.tok_cost[t])));
    b2 = cblock_min(c, b2);
    if (threadIdx.x == 0 && b2 != INFINITY) atomicMin(&cl->best[npar], f2o(b2));
  }
  CT(9)   // epsilon links + best
  if (!cbar(c)) return false;
  CT(10)  // barrier end
  const int n1 = min(ldi(&cl->n_tok), c.max_tok), l1 = min(ldi(&cl->n_link), c.max_link);
  if (c.gtid == 0) {
    c.l.cost_offset[f] = cost_offset;
    c.l.link_off[2 * f + 2] = lem;
    c.l.tok_off[f + 2] = n1;
    c.l.link_off[2 * f + 3] = l1;
  }
  fr.f = f + 1; fr.t0 = n0; fr.t1 = n1; fr.pl0 = l0; fr.pl1 = l1;
  return true;
}

}  // namespace

template <bool ST, bool CP>
__global__ __launch_bounds__(NT) void wfst_cluster_kernel(Graph g, char* state, size_t state_bytes, Opts o, int max_frames,
                                                           int max_tok, int max_link, int hash, int G, int U,
                                                           const float* __restrict__ logp, const int* __restrict__ lens, int T, int C, int stamped) {
  __shared__ float ll[MAX_C], lastp[MAX_C], redf[NT];
  __shared__ int redi[NT], lsh[8], stk_t[CHASE_DEPTH * NT];
  __shared__ float stk_c[CHASE_DEPTH * NT];
  // block b = (k / 8) * 8G + j * 8 + (k % 8): the G members of cluster (utterance) k all have b % 8 == k % 8, i.e. one XCD
  const int b = blockIdx.x, grp = b / (8 * G), r = b % (8 * G);
  const int j = r / 8, u = grp * 8 + (r % 8);
  if (u >= U) return;
  CCtx c;
  c.g = g; c.o = o; c.max_frames = max_frames; c.max_tok = max_tok; c.max_link = max_link; c.hash = hash;
  layout(state + (size_t)u * state_bytes, max_frames, max_tok, max_link, hash, &c.l);
  c.cl = c.l.clu; c.G = G; c.j = j; c.gtid = j * NT + (int)threadIdx.x; c.gthreads = G * NT;
  c.ll = ll; c.redf = redf; c.redi = redi; c.lsh = lsh; c.stk_t = stk_t; c.stk_c = stk_c;
  c.key = c.l.gkey; c.idx = c.l.gidx;
  c.stamped = stamped; c.stamp = 0u;
  if (threadIdx.x < 8) lsh[threadIdx.x] = 0;
  if ((int)threadIdx.x < MAX_C) lastp[threadIdx.x] = c.l.last_prob[threadIdx.x];   // every member keeps its own copy of the remembered blank frame
  __syncthreads();
  Clu* cl = c.cl;
  Hdr* h = c.l.h;
  c.bar_target = cl->bar_base;                 // written before the previous launch ended (kernel boundary: visible)
  // launch prologue: counters from the header, placement check, the newest frame's best cost
  int nf = h->n_frames, num_input = h->num_input, is_last_blank = h->is_last_blank, last_best = h->last_best;
  if (c.gtid == 0) { cl->n_tok = h->n_tok; cl->n_link = h->n_link; cl->overflow = h->overflow; cl->best[nf & 1] = UMAX; cl->cand_min[nf & 1] = UMAX; cl->narcs[nf & 1] = 0; cl->n_heavy = 0; }
  if (threadIdx.x == 0) cl->xcc[j] = (int)xcc_of();
  if (!cbar(c)) return;
  {
    int same = 1;
    for (int m = 1; m < G; ++m) same &= (ldi(&cl->xcc[m]) == ldi(&cl->xcc[0]));
    if (!same) {                               // not behind one L2: plain stores + sc1 loads would not be coherent
      if (c.gtid == 0) { h->overflow |= 64; cl->bar_base = c.bar_target; }
      return;
    }
  }
  CFrame fr;
  fr.f = nf; fr.t0 = c.l.tok_off[nf]; fr.t1 = c.l.tok_off[nf + 1];
  fr.pl0 = fr.pl1 = 0;
  {
    float b0 = INFINITY;
    for (int t = fr.t0 + c.gtid; t < fr.t1; t += c.gthreads) b0 = fminf(b0, o2f(c.l.tok_cost[t]));
    b0 = cblock_min(c, b0);
    if (threadIdx.x == 0 && b0 != INFINITY) atomicMin(&cl->best[nf & 1], f2o(b0));
  }
  bool ok = cbar(c);
#ifdef B2T_WFST_TIMING
  c.tprev = __builtin_amdgcn_s_memtime();
#endif
  const int n = lens ? min(lens[u], T) : T;
  for (int i = 0; i < n && ok; ++i) {
    const float* row = logp + ((size_t)u * T + i) * C;
    // the blank-skipping decision (ctc_wfst_beam_search.cc:70-121) is taken by every member from the same numbers
    int mode = 0;
    const float blank_score = expf(row[0]);
    if (blank_score > o.blank_skip_thresh) {
      is_last_blank = 1;
      __syncthreads();
      if ((int)threadIdx.x < C) lastp[threadIdx.x] = row[threadIdx.x];
    } else {
      int cur_best = 0; float bv = row[0];
      for (int k = 1; k < C; ++k) if (row[k] > bv) { bv = row[k]; cur_best = k; }
      mode = (cur_best != 0 && is_last_blank && cur_best == last_best) ? 2 : 1;
      last_best = cur_best;
    }
    if (mode == 2) {
      __syncthreads();
      if ((int)threadIdx.x < C) ll[threadIdx.x] = o.acoustic_scale * lastp[threadIdx.x];
      if (c.gtid == 0 && fr.f < max_frames) c.l.mapping[fr.f] = num_input - 1;
      __syncthreads();
      ok = cadvance<ST, CP>(c, fr);
    }
    if (mode >= 1 && ok) {
      __syncthreads();
      if ((int)threadIdx.x < C) ll[threadIdx.x] = o.acoustic_scale * row[threadIdx.x];
      if (c.gtid == 0 && fr.f < max_frames) c.l.mapping[fr.f] = num_input;
      __syncthreads();
      ok = cadvance<ST, CP>(c, fr);
      is_last_blank = 0;
    }
    num_input += 1;
  }
  if (ok) {
    cbest_links(c, fr.pl0, fr.pl1);            // the last frame's backpointers
    ok = cbar(c);
  }
#ifdef B2T_WFST_TIMING
  if (c.gtid == 0 && u == 0)
    printf("wfst cluster u0 ticks: other %llu | cutoff %llu | passA light %llu | passA bar+heavy %llu | bestlinks %llu | clears %llu | barA %llu | passB %llu | barB %llu | closure %llu | epslinks %llu | barEnd %llu\n",
           c.tacc[0], c.tacc[1], c.tacc[11], c.tacc[2], c.tacc[3], c.tacc[4], c.tacc[5], c.tacc[6], c.tacc[7], c.tacc[8], c.tacc[9], c.tacc[10]);
  if (c.gtid == 0 && u == 0) printf("wfst cluster u0: heavy tokens %llu over %llu walks; frames %d, tokens %d, links %d, arcs %u\n", c.tacc[12], c.tacc[13], fr.f, ldi(&cl->n_tok), ldi(&cl->n_link), h->arcs_lo);
#endif
  __syncthreads();
  if (c.j == 0 && (int)threadIdx.x < MAX_C) c.l.last_prob[threadIdx.x] = lastp[threadIdx.x];
  if (c.gtid == 0) {
    h->n_frames = fr.f; h->num_input = num_input; h->is_last_blank = is_last_blank; h->last_best = last_best;
    h->n_tok = ldi(&cl->n_tok); h->n_link = ldi(&cl->n_link); h->overflow = ldi(&cl->overflow);
    cl->bar_base = c.bar_target;
  }
}

namespace {
}  // namespace

__global__ __launch_bounds__(NT) void wfst_reset_kernel(Graph g, char* state, size_t state_bytes, Opts o, int max_frames,
                                                         int max_tok, int max_link, int hash, int use_lds) {
  extern __shared__ int dyn[];
  __shared__ float ll[MAX_C], redf[NT];
  __shared__ int redi[NT], sh[8], wl[WL_CAP];
  Ctx c;
  setup(c, g, state, blockIdx.x, state_bytes, o, max_frames, max_tok, max_link, hash, use_lds ? dyn : nullptr, ll, redf, redi, sh, wl);
  init_decoding(c);
}

// CtcWfstBeamSearch::Search (ctc_wfst_beam_search.cc:70-121) over rows [0, lens[u]) of logp[u]
__global__ __launch_bounds__(NT) void wfst_search_kernel(Graph g, char* state, size_t state_bytes, Opts o, int max_frames,
                                                          int max_tok, int max_link, int hash, int use_lds,
                                                          const float* __restrict__ logp, const int* __restrict__ lens, int T, int C) {
  extern __shared__ int dyn[];
  __shared__ float ll[MAX_C], redf[NT];
  __shared__ int redi[NT], sh[8], wl[WL_CAP];
  __shared__ int dec[2];
  Ctx c;
  const int u = blockIdx.x;
  setup(c, g, state, u, state_bytes, o, max_frames, max_tok, max_link, hash, use_lds ? dyn : nullptr, ll, redf, redi, sh, wl);
  if (threadIdx.x == 0) { sh[0] = c.l.h->n_tok; sh[1] = c.l.h->n_link; sh[2] = 0; sh[3] = c.l.h->overflow; }
  __syncthreads();
  const int n = lens ? min(lens[u], T) : T;
  for (int i = 0; i < n; ++i) {
    const float* row = logp + ((size_t)u * T + i) * C;
    if (threadIdx.x == 0) {
      Hdr* h = c.l.h;
      const float blank_score = expf(row[0]);
      int mode = 0;                      // 0: skip the frame, 1: decode it, 2: re-insert the remembered blank frame first
      if (blank_score > o.blank_skip_thresh) {
        h->is_last_blank = 1;
        for (int k = 0; k < C; ++k) c.l.last_prob[k] = row[k];
      } else {
        int cur_best = 0; float bv = row[0];
        for (int k = 1; k < C; ++k) if (row[k] > bv) { bv = row[k]; cur_best = k; }
        mode = (cur_best != 0 && h->is_last_blank && cur_best == h->last_best) ? 2 : 1;
        h->last_best = cur_best;
      }
      dec[0] = mode;
    }
    __syncthreads();
    const int mode = dec[0];
    if (mode == 2) {
      if ((int)threadIdx.x < C) ll[threadIdx.x] = o.acoustic_scale * c.l.last_prob[threadIdx.x];
      if (threadIdx.x == 0 && c.l.h->n_frames < max_frames) c.l.mapping[c.l.h->n_frames] = c.l.h->num_input - 1;
      __syncthreads();
      advance(c);
    }
    if (mode >= 1) {
      if ((int)threadIdx.x < C) ll[threadIdx.x] = o.acoustic_scale * row[threadIdx.x];
      if (threadIdx.x == 0 && c.l.h->n_frames < max_frames) c.l.mapping[c.l.h->n_frames] = c.l.h->num_input;
      __syncthreads();
      advance(c);
      if (threadIdx.x == 0) c.l.h->is_last_blank = 0;
    }
    if (threadIdx.x == 0) c.l.h->num_input += 1;
    __syncthreads();
  }
#ifdef B2T_WFST_TIMING
  if (threadIdx.x == 0 && blockIdx.x == 0)
    printf("wfst u0 cycles: other %llu | cutoff %llu | passA %llu | clear %llu | claim %llu | relax %llu | eps %llu | best_links %llu\n", c.tacc[0], c.tacc[1],
           c.tacc[2], c.tacc[3], c.tacc[4], c.tacc[5], c.tacc[6], c.tacc[7]);
#endif
}

// Best path by backpointers (lattice-faster-online-decoder.cc:58-150): alignment (ilabels), words (olabels), costs.
// use_final: 0 = partial result (any token of the last frame), 1 = with final costs (after b2t_wfst_finalize).
__global__ void wfst_best_path_kernel(Graph g, char* state, size_t state_bytes, int max_frames, int max_tok, int max_link, int hash,
                                      int use_final, int max_len, int* ali, int* ali_frame, int* n_ali, int* words, int* n_words,
                                      float* costs) {
  const int u = blockIdx.x;
  Lay l;
  layout(state + (size_t)u * state_bytes, max_frames, max_tok, max_link, hash, &l);
  const int F = l.h->n_frames;
  if (threadIdx.x == 0) { n_ali[u] = 0; n_words[u] = 0; costs[2 * u] = 0.f; costs[2 * u + 1] = 0.f; }
  if (F == 0) return;
  // the cheapest token of the last frame (first one among equals, as a serial scan finds it): all threads, then a reduction
  __shared__ float r_cost[BP_NT], r_fc[BP_NT];
  __shared__ int r_tok[BP_NT];
  const int t0 = l.tok_off[F], t1 = l.tok_off[F + 1];
  float best = INFINITY, best_fc = 0.f; int bt = -1;
  const bool with_final = use_final && l.h->has_final;
  for (int t = t0 + (int)threadIdx.x; t < t1; t += BP_NT) {
    float cost = o2f(l.tok_cost[t]), fc = 0.f;
    if (with_final) {
      fc = g.final_cost[l.tok_state[t]];
      cost = fc == INFINITY ? INFINITY : cost + fc;
    }
    if (cost < best) { best = cost; bt = t; best_fc = fc; }
  }
  r_cost[threadIdx.x] = best; r_fc[threadIdx.x] = best_fc; r_tok[threadIdx.x] = bt;
  __syncthreads();
  for (int sft = BP_NT / 2; sft > 0; sft >>= 1) {
    if ((int)threadIdx.x < sft) {
      const float oc = r_cost[threadIdx.x + sft]; const int ot = r_tok[threadIdx.x + sft];
      const float mc = r_cost[threadIdx.x]; const int mt = r_tok[threadIdx.x];
      if (ot >= 0 && (mt < 0 || oc < mc || (oc == mc && ot < mt))) {
        r_cost[threadIdx.x] = oc; r_tok[threadIdx.x] = ot; r_fc[threadIdx.x] = r_fc[threadIdx.x + sft];
      }
    }
    __syncthreads();
  }
  best = r_cost[0]; bt = r_tok[0]; best_fc = r_fc[0];
  (void)best;
  if (bt < 0) return;
  // Walk back.  The chain itself -- token -> {its best link, that link's source token} -- is ONE dependent load per hop and is
  // all thread 0 does per hop (the link ids go to LDS); what a link contributes (labels, costs, the frame it belongs to, that
  // frame's cost offset and input frame) is then fetched by all threads at once, frames from a prefix count of the emitting
  // links, and thread 0 only sums and emits from LDS, in walk order (the sums are the serial walk's, bit for bit).  The
  // first version did everything inside the chain: four dependent loads and a branch per hop, ~1.2 us per decoded frame, a
  // quarter of a streamed frame's latency at 100 frames.  Results are written from the end of the buffers, then moved up.
  __shared__ int s_li[BP_CAP], s_il[BP_CAP], s_ol[BP_CAP], s_map[BP_CAP], s_ctl[4], s_cnt[BP_NT];
  __shared__ float s_gc[BP_CAP], s_ac[BP_CAP];
  int na = 0, nw = 0, frame = F - 1;
  float gc = best_fc, ac = 0.f;
  int* a_out = ali + (size_t)u * max_len; int* f_out = ali_frame + (size_t)u * max_len; int* w_out = words + (size_t)u * max_len;
  if (threadIdx.x == 0) { s_ctl[0] = bt; s_ctl[2] = 0; }
  __syncthreads();
  for (;;) {
    if (threadIdx.x == 0) {
      int t = s_ctl[0], n = 0, done = 0;
      while (n < BP_CAP) {
        const long long bw = l.tok_best[t];
        if (bw < 0 || bw == BEST_UNSET) { done = 1; break; }
        s_li[n++] = (int)(bw >> 32);
        t = (int)(unsigned)(bw & 0xffffffffLL);
      }
      s_ctl[0] = t; s_ctl[1] = n; s_ctl[2] = done;
    }
    __syncthreads();
    const int n = s_ctl[1], done = s_ctl[2];
    constexpr int PER = BP_CAP / BP_NT;            // consecutive entries per thread (the prefix count below)
    const int i0 = (int)threadIdx.x * PER;
    int emit = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int i = i0 + k;
      if (i < n) {
        const int li = s_li[i], a = l.link_arc[li];
        const int il = g_il(g, a);
        s_il[i] = il; s_ol[i] = g_ol(g, a); s_gc[i] = l.link_graph[li]; s_ac[i] = l.link_ac[li];
        emit += il != 0;
      }
    }
    s_cnt[threadIdx.x] = emit;
    __syncthreads();
    int before = 0, total = 0;
    for (int w = 0; w < BP_NT; ++w) { const int v = s_cnt[w]; if (w < (int)threadIdx.x) before += v; total += v; }
    {
      int fr = frame - before;
#pragma unroll
      for (int k = 0; k < PER; ++k) {
        const int i = i0 + k;
        if (i < n && s_il[i] != 0) { s_ac[i] -= l.cost_offset[fr]; s_map[i] = l.mapping[fr]; --fr; }
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int i = 0; i < n; ++i) {
        gc += s_gc[i];
        if (s_il[i] != 0) {
          ac += s_ac[i];
          if (na < max_len) { a_out[max_len - 1 - na] = s_il[i]; f_out[max_len - 1 - na] = s_map[i]; }
          ++na;
        }
        if (s_ol[i] != 0) { if (nw < max_len) w_out[max_len - 1 - nw] = s_ol[i]; ++nw; }
      }
      s_ctl[3] = na; s_cnt[0] = nw;
    }
    frame -= total;
    __syncthreads();
    if (done) break;
  }
  na = s_ctl[3]; nw = s_cnt[0];
  const int ka = min(na, max_len), kw = min(nw, max_len);
  // move up: chunk by chunk, a chunk's reads before its writes (its targets never reach a later chunk's sources)
  for (int base = 0; base < max(ka, kw); base += BP_NT) {
    const int i = base + (int)threadIdx.x;
    int va = 0, vf = 0, vw = 0;
    if (i < ka) { va = a_out[max_len - ka + i]; vf = f_out[max_len - ka + i]; }
    if (i < kw) vw = w_out[max_len - kw + i];
    __syncthreads();
    if (i < ka) { a_out[i] = va; f_out[i] = vf; }
    if (i < kw) w_out[i] = vw;
    __syncthreads();
  }
  if (threadIdx.x == 0) { n_ali[u] = ka; n_words[u] = kw; costs[2 * u] = gc; costs[2 * u + 1] = ac; }
}


namespace {
// PruneForwardLinks (:297-374) / PruneForwardLinksFinal (:380-470) for ONE frame f, by one workgroup.
// extra_cost(t) = min over the surviving forward links of t of (extra_cost(dst) + link cost - cost gap), plus, on the last
// frame of a finished utterance, the final-cost term.  The emitting links of f end in frame f + 1, whose values are final:
// ONE pass over them gives each token a base value (and prunes the links beyond lattice_beam).  The epsilon links stay
// inside the frame and form chains a few arcs deep: they are relaxed IN PLACE from above (atomicMin, Bellman-Ford) until
// nothing moves -- a few passes over a few thousand links instead of over all ~25 k links of the frame each time --, and one
// more pass then prunes the epsilon links beyond the beam with the converged values.  (Pruning while the values are still
// upper bounds would remove links that belong in the lattice.)
// keep_all: the frame's tokens are never removed and count with extra cost 0 (the newest frame in PruneActiveTokens).
// Returns through *flags: [0] scratch, [1] |= an extra cost moved by more than delta (vs tok_prev = its old value: the
// reference's extra_costs_changed, which alone sends PruneActiveTokens one frame further back, :528-531), [2] |= a link was pruned.
#ifdef B2T_FIN_TIMING
__device__ unsigned long long fin_t[8];   // [0..4] cycles in: token init, emitting links, epsilon sweeps, epsilon prune, token pass; [5] sweeps; [6] frames
#define FT(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); fin_t[i] += now_ - ft_; ft_ = now_; } } while (0)
#else
#define FT(i) do {} while (0)
#endif
constexpr int PRUNE_LDS_WORDS = 36000;   // dynamic LDS of the finalize / prune kernels (144 000 B of the CU's 160 KB)
// (l, g, o BY VALUE, and the kernels below hold their Lay as a by-value copy: with a reference to the struct the compiler kept
//  all 25 array pointers in scratch memory, reloaded them around every barrier and -- their address space lost on the way
//  through memory -- turned every access into a FLAT instruction, which also waits on the LDS counter)
__device__ __forceinline__ void prune_frame(const Lay l, const Graph g, const Opts o, int f, int F, bool final_frame, int has_final, float final_best,
                            float delta, int* flags, unsigned* lds) {
  const unsigned INF_BITS = 0x7f800000u;
#ifdef B2T_FIN_TIMING
  unsigned long long ft_ = __builtin_amdgcn_s_memtime();
  if (blockIdx.x == 0 && threadIdx.x == 0) ++fin_t[6];
#endif
  const int a0 = l.tok_off[f], a1 = l.tok_off[f + 1];
  const int e0 = f == 0 ? 0 : l.link_off[2 * f], e1 = l.link_off[2 * f + 1];                 // eps links of frame f
  const int m0 = f < F ? l.link_off[2 * f + 1] : 0, m1 = f < F ? l.link_off[2 * f + 2] : 0;   // emitting f -> f+1
  // The pass is bound by ONE CU's rate of random gathers (64 cache lines per wave instruction): per emitting link the costs of
  // its two tokens, the extra cost of its destination, and an atomic on the extra cost of its source (62 % of finalize's
  // time).  A frame's tokens are contiguous, so the three arrays that are hit at random -- extra costs of frame f (atomics),
  // costs and extra costs of frame f + 1 -- are staged in LDS when they fit (a frame holds ~8 k tokens: 100 KB); the source
  // costs stay in memory (links are created token by token: a wave's sources share a few lines).
  const int b0 = a1, b1 = f < F ? l.tok_off[f + 2] : a1;
  const int nA = a1 - a0, nB = b1 - b0;
  if (lds != nullptr && nA + 2 * nB <= PRUNE_LDS_WORDS) {
    unsigned* xA = lds; unsigned* cB = lds + nA; unsigned* xB = cB + nB;
    // (four tokens per thread and trip, a trip's loads before its stores: tok_prev / tok_extra may alias for all the compiler
    //  knows, so a one-token loop waits out a memory round trip per token)
    for (int tb = a0; tb < a1; tb += 4 * NT) {
      unsigned ex[4]; float base[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int t = tb + k * NT + (int)threadIdx.x, q = t < a1 ? t : a1 - 1;
        ex[k] = l.tok_extra[q];
        base[k] = INFINITY;
        if (final_frame) {
          const float fc = has_final ? g.final_cost[l.tok_state[q]] : 0.f;
          base[k] = fmaxf(o2f(l.tok_cost[q]) + fc - final_best, 0.f);
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int t = tb + k * NT + (int)threadIdx.x;
        if (t < a1) { l.tok_prev[t] = ex[k]; xA[t - a0] = __float_as_uint(base[k]); }
      }
    }
    for (int tb = b0; tb < b1; tb += 4 * NT) {
      unsigned c4[4], x4[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) { const int t = tb + k * NT + (int)threadIdx.x, q = t < b1 ? t : b1 - 1; c4[k] = l.tok_cost[q]; x4[k] = l.tok_extra[q]; }
#pragma unroll
      for (int k = 0; k < 4; ++k) { const int t = tb + k * NT + (int)threadIdx.x; if (t < b1) { cB[t - b0] = c4[k]; xB[t - b0] = x4[k]; } }
    }
    __syncthreads();
    FT(0);
    // A frame has ~22 k emitting links (up to 54 k).  What bounds the pass on ONE CU is the address path of the vector memory
    // unit: a wave's load instruction costs it 16 clocks whatever its width, and a link read field by field is six of them
    // (measured 1.5 clocks per link = 6 x 16 / 64).  So a thread takes FOUR CONSECUTIVE links: one 16-byte load per field, the
    // four alive bytes as one word, and only the source costs remain single gathers; dead links are marked by rewriting that
    // word once.  A trip is still two dependent round trips (links, then source costs), and the marks may alias anything for
    // all the compiler knows, so three trips are kept in flight by hand: trip i + 2 loads its links, trip i + 1 gathers its
    // source costs, trip i is evaluated (LDS reads, LDS atomics, marks).
    const int mb = m0 & ~3;                                   // quads are 16-byte aligned in every link array
    const int n_trips = (m1 - mb + 4 * NT - 1) / (4 * NT);
    const int q_last = m1 > mb ? (m1 - 1 - mb) / 4 : 0;
    int4 srcA, dstA; float4 acA, grA; unsigned alA;
    int4 srcB, dstB; float4 acB, grB; unsigned alB; unsigned csB[4];
#define B2T_LOAD_A(trip)                                                                                                  \
    {                                                                                                                       \
      int q_ = (trip) * NT + (int)threadIdx.x; if (q_ > q_last) q_ = q_last;                                                \
      const int i_ = mb + 4 * q_;                                                                                           \
      alA = *reinterpret_cast<const unsigned*>(l.link_alive + i_);                                                          \
      srcA = *reinterpret_cast<const int4*>(l.link_src + i_); dstA = *reinterpret_cast<const int4*>(l.link_dst + i_);       \
      acA = *reinterpret_cast<const float4*>(l.link_ac + i_); grA = *reinterpret_cast<const float4*>(l.link_graph + i_);    \
    }
#define B2T_A_TO_B(trip)                                                                                                  \
    {                                                                                                                       \
      alB = alA; srcB = srcA; dstB = dstA; acB = acA; grB = grA;                                                            \
      const int i_ = mb + 4 * ((trip) * NT + (int)threadIdx.x);                                                             \
      const int s_[4] = {srcA.x, srcA.y, srcA.z, srcA.w};                                                                   \
      _Pragma("unroll") for (int k = 0; k < 4; ++k) csB[k] = (i_ + k >= m0 && i_ + k < m1) ? l.tok_cost[s_[k]] : 0u;        \
    }
    if (n_trips > 0) {
      B2T_LOAD_A(0);
      B2T_A_TO_B(0);
      if (n_trips > 1) { B2T_LOAD_A(1); }
    }
    for (int tr = 0; tr < n_trips; ++tr) {
      const int4 srcC = srcB, dstC = dstB; const float4 acC = acB, grC = grB; const unsigned alC = alB;
      unsigned csC[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) csC[k] = csB[k];
      if (tr + 1 < n_trips) { B2T_A_TO_B(tr + 1); }
      if (tr + 2 < n_trips) { B2T_LOAD_A(tr + 2); }
      const int i0 = mb + 4 * (tr * NT + (int)threadIdx.x);
      const int s4[4] = {srcC.x, srcC.y, srcC.z, srcC.w}, d4[4] = {dstC.x, dstC.y, dstC.z, dstC.w};
      const float a4[4] = {acC.x, acC.y, acC.z, acC.w}, g4[4] = {grC.x, grC.y, grC.z, grC.w};
      unsigned al_new = alC;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int li = i0 + k;
        if (li < m0 || li >= m1 || !((alC >> (8 * k)) & 0xffu)) continue;
        float lec = __uint_as_float(xB[d4[k] - b0]) + ((o2f(csC[k]) + a4[k] + g4[k]) - o2f(cB[d4[k] - b0]));
        if (lec > o.lattice_beam) { al_new &= ~(0xffu << (8 * k)); continue; }
        if (lec < 0.f) lec = 0.f;
        atomicMin(&xA[s4[k] - a0], __float_as_uint(lec));
      }
      // (the word's other bytes, if any, are links of neighbouring segments: finished, or not started before the next barrier)
      if (al_new != alC) { *reinterpret_cast<unsigned*>(l.link_alive + i0) = al_new; flags[2] = 1; }
    }
#undef B2T_LOAD_A
#undef B2T_A_TO_B
    __syncthreads();
    FT(1);
    for (int iter = 0; iter < 4096 && e1 > e0; ++iter) {
      if (threadIdx.x == 0) flags[0] = 0;
      __syncthreads();
      for (int li = e1 - 1 - (int)threadIdx.x; li >= e0; li -= NT) {
        if (!l.link_alive[li]) continue;
        const int src = l.link_src[li], dst = l.link_dst[li];
        const unsigned de = __hip_atomic_load(&xA[dst - a0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        float lec = __uint_as_float(de) + ((o2f(l.tok_cost[src]) + l.link_ac[li] + l.link_graph[li]) - o2f(l.tok_cost[dst]));
        if (!(lec <= o.lattice_beam)) continue;
        if (lec < 0.f) lec = 0.f;
        const unsigned nb = __float_as_uint(lec);
        if (nb < atomicMin(&xA[src - a0], nb)) flags[0] = 1;
      }
      __syncthreads();
#ifdef B2T_FIN_TIMING
      if (blockIdx.x == 0 && threadIdx.x == 0) ++fin_t[5];
#endif
      if (!flags[0]) break;
      __syncthreads();
    }
    FT(2);
    for (int li = e0 + threadIdx.x; li < e1; li += NT) {
      if (!l.link_alive[li]) continue;
      const int src = l.link_src[li], dst = l.link_dst[li];
      const float lec = __uint_as_float(xA[dst - a0]) + ((o2f(l.tok_cost[src]) + l.link_ac[li] + l.link_graph[li]) - o2f(l.tok_cost[dst]));
      if (lec > o.lattice_beam) { l.link_alive[li] = 0; flags[2] = 1; }
    }
    __syncthreads();
    FT(3);
    for (int t = a0 + threadIdx.x; t < a1; t += NT) {
      unsigned nv = xA[t - a0];
      if (final_frame && __uint_as_float(nv) > o.lattice_beam) nv = INF_BITS;
      l.tok_extra[t] = nv;
      const unsigned ov = l.tok_prev[t];
      if (nv != ov && (nv == INF_BITS || ov == INF_BITS || fabsf(__uint_as_float(nv) - __uint_as_float(ov)) > delta)) flags[1] = 1;
    }
    __syncthreads();
    FT(4);
    return;
  }
  for (int t = a0 + threadIdx.x; t < a1; t += NT) {
    float base = INFINITY;
    if (final_frame) {
      const float fc = has_final ? g.final_cost[l.tok_state[t]] : 0.f;
      base = o2f(l.tok_cost[t]) + fc - final_best;
      if (base < 0.f) base = 0.f;
    }
    l.tok_prev[t] = l.tok_extra[t];
    l.tok_extra[t] = __float_as_uint(base);
  }
  __syncthreads();
  FT(0);
  // (4 links per thread and trip, every load of the four issued before the first use: the pass is a chain of dependent
  //  gathers -- link -> its two tokens -> their costs -- and one workgroup has to hide their latency by itself)
  for (int base = m0; base < m1; base += 4 * NT) {
    int li[4], src[4], dst[4]; unsigned char al[4]; float ac[4], gr[4], cs[4], cd[4], xd[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      li[k] = base + k * NT + (int)threadIdx.x;
      const int q = li[k] < m1 ? li[k] : m1 - 1;
      al[k] = l.link_alive[q]; src[k] = l.link_src[q]; dst[k] = l.link_dst[q]; ac[k] = l.link_ac[q]; gr[k] = l.link_graph[q];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { cs[k] = o2f(l.tok_cost[src[k]]); cd[k] = o2f(l.tok_cost[dst[k]]); xd[k] = __uint_as_float(l.tok_extra[dst[k]]); }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (li[k] >= m1 || !al[k]) continue;
      float lec = xd[k] + ((cs[k] + ac[k] + gr[k]) - cd[k]);
      if (lec > o.lattice_beam) { l.link_alive[li[k]] = 0; flags[2] = 1; continue; }
      if (lec < 0.f) lec = 0.f;
      atomicMin(&l.tok_extra[src[k]], __float_as_uint(lec));
    }
  }
  __syncthreads();
  FT(1);
  // (the relaxation sweeps stay one link per thread and trip: batching four links' loads ahead of their updates doubled the
  //  kernel's time -- a sweep then propagates through fewer links of a chain and more sweeps are needed)
  for (int iter = 0; iter < 4096 && e1 > e0; ++iter) {
    if (threadIdx.x == 0) flags[0] = 0;
    __syncthreads();
    // newest links first: the closure appends the links of deeper tokens later, and extra costs flow from a link's destination
    // to its source, so a sweep in creation order needs one pass per level of the closure and a backward sweep about one in all
    for (int li = e1 - 1 - (int)threadIdx.x; li >= e0; li -= NT) {
      if (!l.link_alive[li]) continue;
      const int src = l.link_src[li], dst = l.link_dst[li];
      const unsigned de = __hip_atomic_load(&l.tok_extra[dst], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      float lec = __uint_as_float(de) + ((o2f(l.tok_cost[src]) + l.link_ac[li] + l.link_graph[li]) - o2f(l.tok_cost[dst]));
      if (!(lec <= o.lattice_beam)) continue;        // cannot survive, and cannot lower anything below the beam
      if (lec < 0.f) lec = 0.f;
      const unsigned nb = __float_as_uint(lec);
      if (nb < atomicMin(&l.tok_extra[src], nb)) flags[0] = 1;
    }
    __syncthreads();
#ifdef B2T_FIN_TIMING
    if (blockIdx.x == 0 && threadIdx.x == 0) ++fin_t[5];
#endif
    if (!flags[0]) break;
    __syncthreads();
  }
  FT(2);
  for (int li = e0 + threadIdx.x; li < e1; li += NT) {
    if (!l.link_alive[li]) continue;
    const int src = l.link_src[li], dst = l.link_dst[li];
    const float lec = __uint_as_float(l.tok_extra[dst]) + ((o2f(l.tok_cost[src]) + l.link_ac[li] + l.link_graph[li]) - o2f(l.tok_cost[dst]));
    if (lec > o.lattice_beam) { l.link_alive[li] = 0; flags[2] = 1; }
  }
  __syncthreads();
  FT(3);
  for (int t = a0 + threadIdx.x; t < a1; t += NT) {
    unsigned nv = l.tok_extra[t];
    if (final_frame && __uint_as_float(nv) > o.lattice_beam) { nv = INF_BITS; l.tok_extra[t] = nv; }
    const unsigned ov = l.tok_prev[t];
    if (nv != ov && (nv == INF_BITS || ov == INF_BITS || fabsf(__uint_as_float(nv) - __uint_as_float(ov)) > delta)) flags[1] = 1;
  }
  __syncthreads();
  FT(4);
}
}  // namespace

// FinalizeDecoding (:632-647): PruneForwardLinksFinal on the last frame, then PruneForwardLinks(delta = 0) +
// PruneTokensForFrame backwards.  Marks link_alive; tok_extra = inf for tokens that leave the lattice.
__global__ __launch_bounds__(NT) void wfst_finalize_kernel(Graph g, char* state, size_t state_bytes, Opts o, int max_frames,
                                                            int max_tok, int max_link, int hash) {
  __shared__ float redf[NT];
  extern __shared__ unsigned prune_lds[];          // PRUNE_LDS_WORDS words (prune_frame's staging area)
  const int u = blockIdx.x;
  Lay l;
  layout(state + (size_t)u * state_bytes, max_frames, max_tok, max_link, hash, &l);
  const int F = l.h->n_frames;
  auto bmin = [&](float v) {
    redf[threadIdx.x] = v; __syncthreads();
    for (int s = NT / 2; s > 0; s >>= 1) { if ((int)threadIdx.x < s) redf[threadIdx.x] = fminf(redf[threadIdx.x], redf[threadIdx.x + s]); __syncthreads(); }
    const float r = redf[0]; __syncthreads(); return r;
  };
  // ComputeFinalCosts (:547-590)
  const int t0 = l.tok_off[F], t1 = l.tok_off[F + 1];
  float b = INFINITY, bf = INFINITY;
  for (int t = t0 + threadIdx.x; t < t1; t += NT) {
    const float cst = o2f(l.tok_cost[t]);
    b = fminf(b, cst); bf = fminf(bf, cst + g.final_cost[l.tok_state[t]]);
  }
  b = bmin(b); bf = bmin(bf);
  const int has_final = bf != INFINITY;
  const float final_best = has_final ? bf : b;
  if (threadIdx.x == 0) { l.h->final_best = final_best; l.h->has_final = has_final; l.h->finalized = 1; }
  for (int li = threadIdx.x; li < min(l.h->n_link, max_link); li += NT) l.link_alive[li] = 1;
  __syncthreads();
  __shared__ int flags[3];
  for (int f = F; f >= 0; --f) prune_frame(l, g, o, f, F, f == F, has_final, final_best, 0.f, flags, prune_lds);
#ifdef B2T_FIN_TIMING
  if (blockIdx.x == 0 && threadIdx.x == 0)
    printf("finalize (utterance 0, 100 MHz ticks): tokens-init %llu, emitting %llu, eps sweeps %llu (%llu sweeps), eps prune %llu, tokens %llu, frames %llu\n",
           fin_t[0], fin_t[1], fin_t[2], fin_t[5], fin_t[3], fin_t[4], fin_t[6]);
#endif
}

// FinalizeDecoding by the utterance's CLUSTER (round 5; verdict item 4): the G workgroups that searched the utterance, behind one
// XCD's L2, instead of one workgroup on one of 256 CUs (5.7 ms for 32 utterances of 111 frames, a third of a pipelined batch).
// The same fixpoints as prune_frame -- extra costs are minima, a link lives iff its converged link-extra-cost is within the
// lattice beam -- so link_alive / tok_extra equal the single-workgroup kernel's bit for bit (tested).  What changes is where the
// minima are taken (L2 atomics on tok_extra instead of LDS) and how often the members meet: tokens of ALL frames are initialised
// up front (FinalizeDecoding walks every frame), then per frame ONE cluster barrier behind the emitting links f -> f + 1 (the
// epsilon links of frame f + 1 are pruned in the same phase: their extras have converged) and one per epsilon sweep of frame f
// (a rotating set of 'something moved' words in L2, no barrier to reset one).  Every link's alive byte is written exactly once.
// The final frame's tokens beyond the beam are marked (extra = inf) at the very end; the one phase that must see the marks
// (the emitting links F - 1 -> F) applies the rule on the fly, the phases that must not (epsilon sweeps / prune of frame F) run
// before anything is marked -- the order of PruneForwardLinksFinal (:380-470).
__global__ __launch_bounds__(NT) void wfst_finalize_cluster_kernel(Graph g, char* state, size_t state_bytes, Opts o, int max_frames,
                                                                    int max_tok, int max_link, int hash, int G, int U) {
  __shared__ float redf[NT];
  __shared__ int redi[NT], lsh[8];
  const int b = blockIdx.x, grp = b / (8 * G), r = b % (8 * G);
  const int j = r / 8, u = grp * 8 + (r % 8);          // (the search's mapping: the G members of utterance u share b % 8, i.e. one XCD)
  if (u >= U) return;
  CCtx c;
  c.g = g; c.o = o; c.max_frames = max_frames; c.max_tok = max_tok; c.max_link = max_link; c.hash = hash;
  layout(state + (size_t)u * state_bytes, max_frames, max_tok, max_link, hash, &c.l);
  c.cl = c.l.clu; c.G = G; c.j = j; c.gtid = j * NT + (int)threadIdx.x; c.gthreads = G * NT;
  c.redf = redf; c.redi = redi; c.lsh = lsh;
  c.ll = nullptr; c.key = nullptr; c.idx = nullptr; c.stk_t = nullptr; c.stk_c = nullptr; c.stamped = 0; c.stamp = 0u;
  if (threadIdx.x < 8) lsh[threadIdx.x] = 0;
  __syncthreads();
  const Lay l = c.l;          // a copy, not a reference: see prune_frame
  Clu* cl = c.cl;
  Hdr* h = l.h;
  c.bar_target = cl->bar_base;
  const unsigned INF_BITS = 0x7f800000u;
  const int F = h->n_frames;
  const float beam = o.lattice_beam;
  // ComputeFinalCosts (:547-590): every member reduces the whole last frame itself (a few thousand tokens: cheaper than a barrier)
  const int tF0 = l.tok_off[F], tF1 = min(l.tok_off[F + 1], max_tok);
  float bb = INFINITY, bf = INFINITY;
  for (int t = tF0 + (int)threadIdx.x; t < tF1; t += NT) {
    const float cst = o2f(l.tok_cost[t]);
    bb = fminf(bb, cst); bf = fminf(bf, cst + g.final_cost[l.tok_state[t]]);
  }
  bb = cblock_min(c, bb); bf = cblock_min(c, bf);
  const int has_final = bf != INFINITY;
  const float final_best = has_final ? bf : bb;
  if (c.gtid == 0) {
    cl->overflow = h->overflow;
    for (int k = 0; k < 8; ++k) cl->changed[k] = 0;
  }
  // extra costs of every frame's tokens: inf, the last frame's from the final costs
  for (int t = c.gtid; t < tF1; t += c.gthreads) {
    unsigned v = INF_BITS;
    if (t >= tF0) {
      const float fc = has_final ? g.final_cost[l.tok_state[t]] : 0.f;
      v = __float_as_uint(fmaxf(o2f(l.tok_cost[t]) + fc - final_best, 0.f));
    }
    l.tok_extra[t] = v;
  }
  bool ok = cbar(c);
  int it = 0;                  // epsilon sweeps so far (all frames): sweep `it` reports through cl->changed[it & 7]
  for (int f = F; f >= 0 && ok; --f) {
    // ---- emitting links f -> f + 1 (their destinations' extras are final) ...
    if (f < F) {
      const int m0 = l.link_off[2 * f + 1], m1 = min(l.link_off[2 * f + 2], max_link);
      const bool into_last = f + 1 == F;
      for (int li = m0 + c.gtid; li < m1; li += c.gthreads) {
        const int src = l.link_src[li], dst = l.link_dst[li];
        float xd = __uint_as_float(ldu(&l.tok_extra[dst]));
        if (into_last && xd > beam) xd = INFINITY;          // (the mark PruneForwardLinksFinal has left on the last frame by now)
        float lec = xd + ((o2f(l.tok_cost[src]) + l.link_ac[li] + l.link_graph[li]) - o2f(l.tok_cost[dst]));
        if (lec > beam) { l.link_alive[li] = 0; continue; }
        l.link_alive[li] = 1;
        if (lec < 0.f) lec = 0.f;
        atomicMin(&l.tok_extra[src], __float_as_uint(lec));
      }
      // ... and the epsilon links of frame f + 1, whose sweeps have converged: pruned against the (unmarked) extras
      const int q0 = l.link_off[2 * (f + 1)], q1 = min(l.link_off[2 * (f + 1) + 1], max_link);
      for (int li = q0 + c.gtid; li < q1; li += c.gthreads) {
        const int src = l.link_src[li], dst = l.link_dst[li];
        const float lec = __uint_as_float(ldu(&l.tok_extra[dst])) + ((o2f(l.tok_cost[src]) + l.link_ac[li] + l.link_graph[li]) - o2f(l.tok_cost[dst]));
        l.link_alive[li] = lec > beam ? 0 : 1;
      }
      ok = cbar(c);
      if (!ok) break;
    }
    // ---- epsilon links inside frame f: relaxation sweeps (newest links first, as prune_frame) until nothing moves
    const int e0 = f == 0 ? 0 : l.link_off[2 * f], e1 = min(l.link_off[2 * f + 1], max_link);
    for (int iter = 0; iter < 4096 && e1 > e0; ++iter) {
      const int w = it & 7;
      if (c.gtid == 0) cl->changed[(it + 1) & 7] = 0;       // (last read seven sweeps ago; the next sweep writes it behind this sweep's barrier)
      int moved = 0;
      // (two relaxation sweeps per meeting: the atomics of one member are in L2 for the others' next loads at once, so values travel
      //  two links further per barrier; a round in which NOBODY lowered anything read only final values: the fixpoint)
      for (int rep = 0; rep < 2; ++rep)
      for (int li = e1 - 1 - c.gtid; li >= e0; li -= c.gthreads) {
        const int src = l.link_src[li], dst = l.link_dst[li];
        float lec = __uint_as_float(ldu(&l.tok_extra[dst])) + ((o2f(l.tok_cost[src]) + l.link_ac[li] + l.link_graph[li]) - o2f(l.tok_cost[dst]));
        if (!(lec <= beam)) continue;
        if (lec < 0.f) lec = 0.f;
        const unsigned nb = __float_as_uint(lec);
        if (nb < atomicMin(&l.tok_extra[src], nb)) moved = 1;
      }
      if (moved) cl->changed[w] = 1;
      ok = cbar(c);
      ++it;
      if (!ok || !ldi(&cl->changed[w])) break;
    }
    if (!ok) break;
  }
  if (ok) {
    // epsilon links of frame 0, then the marks on the last frame (nothing reads its extras any more)
    const int q1 = min(l.link_off[1], max_link);
    for (int li = c.gtid; li < q1; li += c.gthreads) {
      const int src = l.link_src[li], dst = l.link_dst[li];
      const float lec = __uint_as_float(ldu(&l.tok_extra[dst])) + ((o2f(l.tok_cost[src]) + l.link_ac[li] + l.link_graph[li]) - o2f(l.tok_cost[dst]));
      l.link_alive[li] = lec > beam ? 0 : 1;
    }
    ok = cbar(c);                // (frame 0 may BE the last frame: its prune reads the unmarked extras)
    if (ok)
      for (int t = tF0 + c.gtid; t < tF1; t += c.gthreads)
        if (__uint_as_float(ldu(&l.tok_extra[t])) > beam) l.tok_extra[t] = INF_BITS;
  }
  __syncthreads();
  if (c.gtid == 0) {
    h->final_best = final_best; h->has_final = has_final; h->finalized = 1;
    h->overflow = ldi(&cl->overflow);
    cl->bar_base = c.bar_target;
  }
}

// PruneActiveTokens (lattice-faster-decoder.cc:516-545, called every prune_interval frames at :592-630) as a pass of its own
// between two search calls: PruneForwardLinks (:297-374) on the frames F-1 .. 0 -- the tokens of the newest frame F are
// never pruned and count with extra_cost 0 --, going back only as far as something still changes, then PruneTokensForFrame
// (:489-514) as a stable in-place COMPACTION of the token and link arrays (the reference frees list nodes; here the arrays
// of the state block shrink, so a streamed utterance holds its pruned lattice plus at most prune_interval raw frames).
// Extra costs computed against the best path SO FAR are lower bounds of the final ones, so this removes only what
// FinalizeDecoding would remove: the final lattice is the same with or without these passes (tested).
__global__ __launch_bounds__(NT) void wfst_prune_kernel(Graph g, char* state, size_t state_bytes, Opts o, int max_frames,
                                                         int max_tok, int max_link, int hash, float delta, float min_fill) {
  __shared__ float redf[NT];
  extern __shared__ unsigned prune_lds[];          // PRUNE_LDS_WORDS words (prune_frame's staging area)
  const int u = blockIdx.x;
  Lay l0;
  layout(state + (size_t)u * state_bytes, max_frames, max_tok, max_link, hash, &l0);
  const Lay l = l0;         // a copy, not a reference: see prune_frame
  const int F = l.h->n_frames;
  if (F < 2 || l.h->overflow || l.h->finalized) return;
  // memory-pressure policy (min_fill > 0): the pass only exists to bound memory, so an utterance whose arrays are still
  // mostly empty skips it (min_fill = 0: every call prunes, the reference's fixed prune_interval)
  if ((float)l.h->n_tok < min_fill * (float)max_tok && (float)l.h->n_link < min_fill * (float)max_link) return;
  const unsigned INF_BITS = 0x7f800000u;
  const int n_tok = min(l.h->n_tok, max_tok), n_link = min(l.h->n_link, max_link);
  for (int li = l.h->links_marked + (int)threadIdx.x; li < n_link; li += NT) l.link_alive[li] = 1;
  __syncthreads();
  // ---- PruneForwardLinks, frames F-1 .. 0, stopping at the first frame where nothing moved by more than delta
  __shared__ int flags[3];
#ifdef B2T_WFST_TIMING
  unsigned long long tp0 = __builtin_amdgcn_s_memtime(), tp1, tp2, tp3, tp4;
#endif
  int f_stop = -1;
  for (int f = F - 1; f >= 0; --f) {
    __syncthreads();
    if (threadIdx.x == 0) flags[1] = 0;
    __syncthreads();
    prune_frame(l, g, o, f, F, false, 0, 0.f, delta, flags, prune_lds);
    if (!flags[1]) { f_stop = f; break; }
  }
  // ---- compaction of the tokens of frames f_stop+1 .. F-1 and of every link that starts in frame f_stop or later.
  // Stable and in place, one array SEGMENT at a time (a frame's tokens; a frame's epsilon links; its emitting links), each in
  // chunks of 4 x NT elements: 4 flags per thread, ONE barrier per chunk (wave scans + the 4 x 16 wave totals read by
  // everyone from a double-buffered LDS table; the same barrier separates the chunk's reads from its writes).  The first
  // version scanned NT elements per chunk with four barriers and a serial boundary loop: 18 ms per pass, mostly barriers.
#ifdef B2T_WFST_TIMING
  tp1 = __builtin_amdgcn_s_memtime();
#endif
  const int T0 = l.tok_off[f_stop + 1];
  const int TF = l.tok_off[F];                      // tokens of the newest frame always stay
  // (epsilon links of the stop frame that were pruned just now stay behind as dead entries -- link_alive 0 --: the stop frame's
  //  tokens keep their ids and their backpointers into that range; FinalizeDecoding prunes them again)
  const int L0 = f_stop >= 0 ? l.link_off[2 * f_stop + 1] : 0;
  int* excl = reinterpret_cast<int*>(l.tok_prev);   // [t] = new id of token t (or -1)
  __shared__ int wtot[2][4][NT / 64];
  int flip = 0;
  // exclusive positions of 4 flags per thread (element e_k = base + k * NT + tid: coalesced) + the chunk's total
  auto scan4 = [&](const int (&fl)[4], int (&pos)[4], int& total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int incl[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      int v = fl[k];
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) { const int x = __shfl_up(v, off, 64); if (lane >= off) v += x; }
      incl[k] = v;
      if (lane == 63) wtot[flip][k][w] = v;
    }
    __syncthreads();
    int run = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      int before = 0, row = 0;
      for (int ww = 0; ww < NT / 64; ++ww) { const int v = wtot[flip][k][ww]; if (ww < w) before += v; row += v; }
      pos[k] = run + before + incl[k] - fl[k];
      run += row;
    }
    total = run;
    flip ^= 1;
  };
  auto tok_alive = [&](int t) { return t >= TF || l.tok_extra[t] != INF_BITS; };
  // pass 1: new token ids frame by frame; tok_off rewritten as the frames are finished
  int run_t = T0;
  {
    int seg0 = T0;
    for (int fb = f_stop + 1; fb <= F; ++fb) {
      const int seg1 = min(l.tok_off[fb + 1], n_tok);       // (old value: rewritten below, after everyone has read it)
      for (int base = seg0; base < seg1; base += 4 * NT) {
        int fl[4], pos[4], tot;
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int t = base + k * NT + (int)threadIdx.x; fl[k] = t < seg1 ? (int)tok_alive(t) : 0; }
        scan4(fl, pos, tot);
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int t = base + k * NT + (int)threadIdx.x; if (t < seg1) excl[t] = fl[k] ? run_t + pos[k] : -1; }
        run_t += tot;
      }
      __syncthreads();
      if (threadIdx.x == 0) l.tok_off[fb + 1] = run_t;
      seg0 = seg1;
    }
  }
  const int n_tok_new = run_t;
  __syncthreads();
#ifdef B2T_WFST_TIMING
  tp2 = __builtin_amdgcn_s_memtime();
#endif
  // pass 2: links -- drop, remap, move; link_off rewritten segment by segment
  int run_l = L0;
  {
    int seg0 = L0;
    for (int jb = (f_stop >= 0 ? 2 * f_stop + 1 : 0); jb <= 2 * F; ++jb) {
      const int seg1 = min(l.link_off[jb + 1], n_link);
      for (int base = seg0; base < seg1; base += 4 * NT) {
        int fl[4], pos[4], tot, src[4], dst[4], arc[4]; float ac[4], gr[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int li = base + k * NT + (int)threadIdx.x;
          fl[k] = 0;
          if (li < seg1) {
            src[k] = l.link_src[li]; dst[k] = l.link_dst[li]; arc[k] = l.link_arc[li]; ac[k] = l.link_ac[li]; gr[k] = l.link_graph[li];
            if (l.link_alive[li]) {
              if (src[k] >= T0) src[k] = excl[src[k]];
              if (dst[k] >= T0) dst[k] = excl[dst[k]];
              fl[k] = src[k] >= 0 && dst[k] >= 0;
            }
          }
        }
        scan4(fl, pos, tot);                               // (its barrier: every read of this chunk is done)
#pragma unroll
        for (int k = 0; k < 4; ++k) if (fl[k]) {
          const int q = run_l + pos[k];                    // q <= li
          l.link_src[q] = src[k]; l.link_dst[q] = dst[k]; l.link_arc[q] = arc[k]; l.link_ac[q] = ac[k]; l.link_graph[q] = gr[k]; l.link_alive[q] = 1;
        }
        run_l += tot;
      }
      __syncthreads();
      if (threadIdx.x == 0) l.link_off[jb + 1] = run_l;
      seg0 = seg1;
    }
  }
  const int n_link_new = run_l;
  __syncthreads();
#ifdef B2T_WFST_TIMING
  tp3 = __builtin_amdgcn_s_memtime();
#endif
  // pass 3: move the surviving tokens (ids only go down; a chunk's reads are done before its writes)
  for (int base = T0; base < n_tok; base += 4 * NT) {
    int k2[4], st[4]; unsigned cs[4], ex[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int t = base + k * NT + (int)threadIdx.x;
      k2[k] = -1;
      if (t < n_tok) { k2[k] = excl[t]; st[k] = l.tok_state[t]; cs[k] = l.tok_cost[t]; ex[k] = l.tok_extra[t]; }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) if (k2[k] >= 0) { l.tok_state[k2[k]] = st[k]; l.tok_cost[k2[k]] = cs[k]; l.tok_extra[k2[k]] = ex[k]; l.tok_best[k2[k]] = BEST_UNSET; }
    __syncthreads();
  }
  // backpointers of the moved tokens: the first surviving link whose cost equals the token's (best_links' rule)
  for (int li = L0 + (int)threadIdx.x; li < n_link_new; li += NT) {
    const int src = l.link_src[li], dst = l.link_dst[li];
    if (dst < T0) continue;
    const float tot = o2f(l.tok_cost[src]) + l.link_ac[li] + l.link_graph[li];
    if (f2o(tot) == l.tok_cost[dst]) atomicMin(&l.tok_best[dst], best_word(li, src));
  }
  __syncthreads();
#ifdef B2T_WFST_TIMING
  tp4 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0 && u == 0) printf("wfst prune u0: F %d f_stop %d | sweeps %llu | tok ids %llu | links %llu | tok move + best %llu | tokens %d -> %d, links %d -> %d\n",
                                         F, f_stop, tp1 - tp0, tp2 - tp1, tp3 - tp2, tp4 - tp3, n_tok, n_tok_new, n_link, n_link_new);
#endif
  if (threadIdx.x == 0) {
    if (T0 == 0) l.tok_best[0] = -1;
    Hdr* h = l.h;
    h->peak_tok = max(h->peak_tok, n_tok); h->peak_link = max(h->peak_link, n_link);
    h->removed_tok += n_tok - n_tok_new; h->removed_link += n_link - n_link_new;
    h->n_tok = n_tok_new; h->n_link = n_link_new; h->links_marked = n_link_new; h->n_prunes += 1;
  }
}

// PruneActiveTokens by the utterance's CLUSTER (round 5; verdict item 4).  The one-workgroup pass above takes 3.9 ms for 32 utterances
// (more than the 25 frames of search between two passes): ~30 frames of prune_frame on one CU, then a stable compaction that meets
// at a barrier every 4096 elements.  Here the G workgroups that search the utterance share the pass:
//   * PruneForwardLinks per frame as in wfst_finalize_cluster_kernel (L2 atomics on the extra costs), with the pass's own rules:
//     a frame's tokens are re-initialised when its turn comes (the walk stops at the first frame where nothing moved by more than
//     delta), so per frame: emitting links | barrier | epsilon sweeps (one barrier each) | epsilon prune + the 'moved' test +
//     the NEXT frame's initialisation | barrier.  The speculative initialisation of frame f_stop - 1 is undone when the walk stops.
//   * the compaction meets once per 8 x 4096 elements: every member scans its 4096-element share, publishes one total, and after the
//     barrier knows its base; tokens get their new ids from per-member prefix counts (one barrier for the whole range).
// Same surviving set, same order, same ids as the one-workgroup pass (tested array by array on the same state).
// Scratch: the epsilon work list (rebuilt by every frame of the search) for flags and totals; old frame offsets in LDS.
__device__ __forceinline__ unsigned ldub(const unsigned char* p) { return (unsigned)__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__global__ __launch_bounds__(NT) void wfst_prune_cluster_kernel(Graph g, char* state, size_t state_bytes, Opts o, int max_frames,
                                                                 int max_tok, int max_link, int hash, float delta, float min_fill, int G, int U) {
  __shared__ int lsh[8], wtot[2][4][NT / 64], tots[80];     // tots[0 .. G]: token bases; tots[40 .. 40 + G]: a link chunk's bases
  extern __shared__ int old_off[];                 // [max_frames + 3] tok_off, then [2 (max_frames + 3)] link_off, as the pass found them
  const int b = blockIdx.x, grp = b / (8 * G), r = b % (8 * G);
  const int j = r / 8, u = grp * 8 + (r % 8);
  if (u >= U) return;
  CCtx c;
  c.g = g; c.o = o; c.max_frames = max_frames; c.max_tok = max_tok; c.max_link = max_link; c.hash = hash;
  layout(state + (size_t)u * state_bytes, max_frames, max_tok, max_link, hash, &c.l);
  c.cl = c.l.clu; c.G = G; c.j = j; c.gtid = j * NT + (int)threadIdx.x; c.gthreads = G * NT;
  c.redf = nullptr; c.redi = nullptr; c.lsh = lsh;
  c.ll = nullptr; c.key = nullptr; c.idx = nullptr; c.stk_t = nullptr; c.stk_c = nullptr; c.stamped = 0; c.stamp = 0u;
  if (threadIdx.x < 8) lsh[threadIdx.x] = 0;
  __syncthreads();
  const Lay l = c.l;
  Clu* cl = c.cl;
  Hdr* h = l.h;
  c.bar_target = cl->bar_base;
  const int F = h->n_frames;
  // (every member reads the same header, written by the previous launch: the same decision, before any barrier)
  if (F < 2 || h->overflow || h->finalized) return;
  if ((float)h->n_tok < min_fill * (float)max_tok && (float)h->n_link < min_fill * (float)max_link) return;
  const unsigned INF_BITS = 0x7f800000u;
  const float beam = o.lattice_beam;
  const int n_tok = min(h->n_tok, max_tok), n_link = min(h->n_link, max_link);
  int* scr = l.wlg;                                  // [0, 8): 'moved' per frame (mod 8); [64 + 32 (chunk mod 8) + member]: chunk totals; [512 + member]: token totals
  int* tok_off_old = old_off;
  int* link_off_old = old_off + (max_frames + 3);
  for (int i = threadIdx.x; i <= F + 1; i += NT) tok_off_old[i] = l.tok_off[i];
  for (int i = threadIdx.x; i <= 2 * F + 2; i += NT) link_off_old[i] = l.link_off[i];
  if (c.gtid == 0) {
    cl->overflow = h->overflow;
    for (int k = 0; k < 8; ++k) { cl->changed[k] = 0; scr[k] = 0; }
  }
  for (int li = h->links_marked + c.gtid; li < n_link; li += c.gthreads) l.link_alive[li] = 1;
  auto init_frame = [&](int f) {                     // tok_prev = the old extra cost, extra = inf (prune_frame's first pass)
    const int a0 = tok_off_old[f], a1 = tok_off_old[f + 1];
    for (int t = a0 + c.gtid; t < a1; t += c.gthreads) { l.tok_prev[t] = ldu(&l.tok_extra[t]); l.tok_extra[t] = INF_BITS; }
  };
  __syncthreads();
  init_frame(F - 1);
  bool ok = cbar(c);
  int it = 0, f_stop = -1;
  for (int f = F - 1; f >= 0 && ok; --f) {
    const int a0 = tok_off_old[f], a1 = tok_off_old[f + 1];
    {   // emitting links f -> f + 1
      const int m0 = link_off_old[2 * f + 1], m1 = min(link_off_old[2 * f + 2], max_link);
      for (int li = m0 + c.gtid; li < m1; li += c.gthreads) {
        if (!ldub(&l.link_alive[li])) continue;
        const int src = l.link_src[li], dst = l.link_dst[li];
        float lec = __uint_as_float(ldu(&l.tok_extra[dst])) + ((o2f(l.tok_cost[src]) + l.link_ac[li] + l.link_graph[li]) - o2f(l.tok_cost[dst]));
        if (lec > beam) { l.link_alive[li] = 0; continue; }
        if (lec < 0.f) lec = 0.f;
        atomicMin(&l.tok_extra[src], __float_as_uint(lec));
      }
    }
    ok = cbar(c);
    if (!ok) break;
    const int e0 = f == 0 ? 0 : link_off_old[2 * f], e1 = min(link_off_old[2 * f + 1], max_link);
    for (int iter = 0; iter < 4096 && e1 > e0; ++iter) {
      const int w = it & 7;
      if (c.gtid == 0) cl->changed[(it + 1) & 7] = 0;
      int moved = 0;
      for (int rep = 0; rep < 2; ++rep)                  // (two sweeps per meeting: see wfst_finalize_cluster_kernel)
      for (int li = e1 - 1 - c.gtid; li >= e0; li -= c.gthreads) {
        if (!ldub(&l.link_alive[li])) continue;
        const int src = l.link_src[li], dst = l.link_dst[li];
        float lec = __uint_as_float(ldu(&l.tok_extra[dst])) + ((o2f(l.tok_cost[src]) + l.link_ac[li] + l.link_graph[li]) - o2f(l.tok_cost[dst]));
        if (!(lec <= beam)) continue;
        if (lec < 0.f) lec = 0.f;
        const unsigned nb = __float_as_uint(lec);
        if (nb < atomicMin(&l.tok_extra[src], nb)) moved = 1;
      }
      if (moved) cl->changed[w] = 1;
      ok = cbar(c);
      ++it;
      if (!ok || !ldi(&cl->changed[w])) break;
    }
    if (!ok) break;
    // epsilon links beyond the beam; did an extra cost of this frame move by more than delta?; the next frame's initialisation
    for (int li = e0 + c.gtid; li < e1; li += c.gthreads) {
      if (!ldub(&l.link_alive[li])) continue;
      const int src = l.link_src[li], dst = l.link_dst[li];
      const float lec = __uint_as_float(ldu(&l.tok_extra[dst])) + ((o2f(l.tok_cost[src]) + l.link_ac[li] + l.link_graph[li]) - o2f(l.tok_cost[dst]));
      if (lec > beam) l.link_alive[li] = 0;
    }
    {
      int mv = 0;
      for (int t = a0 + c.gtid; t < a1; t += c.gthreads) {
        const unsigned nv = ldu(&l.tok_extra[t]), ov = ldu(&l.tok_prev[t]);
        if (nv != ov && (nv == INF_BITS || ov == INF_BITS || fabsf(__uint_as_float(nv) - __uint_as_float(ov)) > delta)) mv = 1;
      }
      if (mv) scr[f & 7] = 1;
      if (c.gtid == 0) scr[(f + 6) & 7] = 0;           // the word of frame f - 2 (last read two frames ago, by frame f + 6's test)
    }
    if (f > 0) init_frame(f - 1);
    ok = cbar(c);
    if (!ok) break;
    if (!ldi(&scr[f & 7])) { f_stop = f; break; }
  }
  if (ok && f_stop > 0) {                              // the walk stopped: frame f_stop - 1 keeps its old extra costs
    const int a0 = tok_off_old[f_stop - 1], a1 = tok_off_old[f_stop];
    for (int t = a0 + c.gtid; t < a1; t += c.gthreads) l.tok_extra[t] = ldu(&l.tok_prev[t]);
  }
  // ---- compaction: tokens of frames f_stop + 1 .. F - 1 (the newest frame's always stay), links from frame f_stop's emitting ones on
  const int T0 = tok_off_old[f_stop + 1], TF = tok_off_old[F];
  const int L0 = f_stop >= 0 ? link_off_old[2 * f_stop + 1] : 0;
  int* excl = reinterpret_cast<int*>(l.tok_prev);      // alive: number of survivors before t in its member's block; dead: the complement of that
  int flip = 0;
  auto scan4 = [&](const int (&fl)[4], int (&pos)[4], int& total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int incl[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      int v = fl[k];
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) { const int x = __shfl_up(v, off, 64); if (lane >= off) v += x; }
      incl[k] = v;
      if (lane == 63) wtot[flip][k][w] = v;
    }
    __syncthreads();
    int run = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      int before = 0, row = 0;
      for (int ww = 0; ww < NT / 64; ++ww) { const int v = wtot[flip][k][ww]; if (ww < w) before += v; row += v; }
      pos[k] = run + before + incl[k] - fl[k];
      run += row;
    }
    total = run;
    flip ^= 1;
  };
  // tokens: member j scans the j-th of G equal blocks of [T0, n_tok)
  const int ntok_span = n_tok - T0;
  const int tblk = max(4, ((ntok_span + G - 1) / G + 3) & ~3);
  int n_tok_new = T0;
  if (ok) {
    ok = cbar(c);                                      // the undo's loads of tok_prev are done everywhere before tok_prev becomes excl; the last marks are in L2
  }
  if (ok) {
    const int tb0 = T0 + j * tblk, tb1 = min(tb0 + tblk, n_tok);
    int run = 0;
    for (int base = tb0; base < tb1; base += 4 * NT) {
      int fl[4], pos[4], tot;
#pragma unroll
      for (int k = 0; k < 4; ++k) { const int t = base + k * NT + (int)threadIdx.x; fl[k] = t < tb1 ? (int)(t >= TF || ldu(&l.tok_extra[t]) != INF_BITS) : 0; }
      scan4(fl, pos, tot);
#pragma unroll
      for (int k = 0; k < 4; ++k) { const int t = base + k * NT + (int)threadIdx.x; if (t < tb1) excl[t] = fl[k] ? run + pos[k] : ~(run + pos[k]); }
      run += tot;
    }
    if (threadIdx.x == 0) scr[512 + j] = run;
    ok = cbar(c);
  }
  if (ok) {
    if ((int)threadIdx.x <= G) {                       // tots[m] = survivors in the blocks before member m; tots[G] = all
      int sum = 0;
      for (int m = 0; m < (int)threadIdx.x; ++m) sum += ldi(&scr[512 + m]);
      tots[threadIdx.x] = sum;
    }
    __syncthreads();
    n_tok_new = T0 + tots[G];
  }
  auto count_before = [&](int p) {                     // survivors in [T0, p)
    if (p >= n_tok) return tots[G];
    const int v = ldi(&excl[p]);
    return tots[(p - T0) / tblk] + (v < 0 ? ~v : v);
  };
  auto new_id = [&](int t) {                           // t >= T0: its id after the pass, or -1
    const int v = ldi(&excl[t]);
    return v < 0 ? -1 : T0 + tots[(t - T0) / tblk] + v;
  };
  if (ok) {
    for (int fb = f_stop + 1 + c.gtid; fb <= F; fb += c.gthreads) l.tok_off[fb + 1] = T0 + count_before(min(tok_off_old[fb + 1], n_tok));
  }
  // links: chunks of G x 4 NT over [L0, n_link), member j takes the j-th 4 NT of a chunk; ONE barrier per chunk.  The segment offsets
  // (a frame's epsilon links, its emitting links) come out on the way: link_off[jb + 1] = L0 + survivors before its old value p, which
  // the thread holding element p knows once the members' totals are in (the first version walked segment by segment: a barrier per
  // segment, ~60 of the pass's ~75 compaction barriers)
  int run_l = L0, cc = 0;
  if (ok) {
    int jb = f_stop >= 0 ? 2 * f_stop + 1 : 0;         // the next segment end to place (offsets are sorted)
    for (int cb = L0; cb < n_link && ok; cb += G * 4 * NT, ++cc) {
      const int base = cb + j * 4 * NT;
      int fl[4], pos[4], tot, src[4], dst[4], arc[4]; float ac[4], gr[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int li = base + k * NT + (int)threadIdx.x;
        fl[k] = 0;
        if (li < n_link) {
          src[k] = l.link_src[li]; dst[k] = l.link_dst[li]; arc[k] = l.link_arc[li]; ac[k] = l.link_ac[li]; gr[k] = l.link_graph[li];
          if (ldub(&l.link_alive[li])) {
            if (src[k] >= T0) src[k] = new_id(src[k]);
            if (dst[k] >= T0) dst[k] = new_id(dst[k]);
            fl[k] = src[k] >= 0 && dst[k] >= 0;
          }
        }
      }
      scan4(fl, pos, tot);
      if (threadIdx.x == 0) scr[64 + 32 * (cc & 7) + j] = tot;
      ok = cbar(c);                                    // every read of this chunk is done, every member's total is out
      if (!ok) break;
      if ((int)threadIdx.x <= G) {
        int sum = 0;
        for (int m = 0; m < (int)threadIdx.x; ++m) sum += ldi(&scr[64 + 32 * (cc & 7) + m]);
        tots[40 + threadIdx.x] = sum;                  // (tots[0 .. G] keep the token bases)
      }
      __syncthreads();
      const int mybase = run_l + tots[40 + j];
#pragma unroll
      for (int k = 0; k < 4; ++k) if (fl[k]) {
        const int q = mybase + pos[k];                 // q <= li
        l.link_src[q] = src[k]; l.link_dst[q] = dst[k]; l.link_arc[q] = arc[k]; l.link_ac[q] = ac[k]; l.link_graph[q] = gr[k]; l.link_alive[q] = 1;
      }
      // segment ends inside this chunk
      const int cend = cb + G * 4 * NT;
      while (jb <= 2 * F && link_off_old[jb + 1] < cend && link_off_old[jb + 1] < n_link) {
        const int p = link_off_old[jb + 1], e = p - base;          // (p >= cb: the ends are sorted and the earlier ones are placed)
        if (e >= 0 && e < 4 * NT && (e & (NT - 1)) == (int)threadIdx.x) {
          const int k = e / NT;
          l.link_off[jb + 1] = mybase + (k == 0 ? pos[0] : k == 1 ? pos[1] : k == 2 ? pos[2] : pos[3]);
        }
        ++jb;
      }
      run_l += tots[40 + G];
      __syncthreads();                                 // tots[40 ..] are rewritten by the next chunk
    }
    if (ok) for (int q = jb + c.gtid; q <= 2 * F; q += c.gthreads) l.link_off[q + 1] = run_l;      // ends at (or clamped to) the last link
  }
  const int n_link_new = run_l;
  // tokens move down: chunks of G x 4 NT, reads and writes of a chunk separated by a barrier (ids only go down)
  if (ok) {
    for (int cb = T0; cb < n_tok && ok; cb += G * 4 * NT) {
      const int base = cb + j * 4 * NT;
      int k2[4], st[4]; unsigned cs[4], ex[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int t = base + k * NT + (int)threadIdx.x;
        k2[k] = -1;
        if (t < n_tok) { k2[k] = new_id(t); st[k] = l.tok_state[t]; cs[k] = l.tok_cost[t]; ex[k] = ldu(&l.tok_extra[t]); }
      }
      ok = cbar(c);
      if (!ok) break;
#pragma unroll
      for (int k = 0; k < 4; ++k) if (k2[k] >= 0) { l.tok_state[k2[k]] = st[k]; l.tok_cost[k2[k]] = cs[k]; l.tok_extra[k2[k]] = ex[k]; l.tok_best[k2[k]] = BEST_UNSET; }
    }
  }
  if (ok) ok = cbar(c);
  if (ok) {
    // backpointers of the moved tokens: the first surviving link whose cost equals the token's (best_links' rule)
    for (int li = L0 + c.gtid; li < n_link_new; li += c.gthreads) {
      const int src = ldi(&l.link_src[li]), dst = ldi(&l.link_dst[li]);
      if (dst < T0) continue;
      const float tot = o2f(ldu(&l.tok_cost[src])) + ldf(&l.link_ac[li]) + ldf(&l.link_graph[li]);
      if (f2o(tot) == ldu(&l.tok_cost[dst])) atomicMin(&l.tok_best[dst], best_word(li, src));
    }
  }
  __syncthreads();
  if (c.gtid == 0) {
    if (ok) {
      if (T0 == 0) l.tok_best[0] = -1;
      h->peak_tok = max(h->peak_tok, n_tok); h->peak_link = max(h->peak_link, n_link);
      h->removed_tok += n_tok - n_tok_new; h->removed_link += n_link - n_link_new;
      h->n_tok = n_tok_new; h->n_link = n_link_new; h->links_marked = n_link_new; h->n_prunes += 1;
    }
    h->overflow = ldi(&cl->overflow);
    cl->bar_base = c.bar_target;
  }
}

// The pruned lattice in compact form (GetRawLattice, lattice-faster-decoder.cc:106-186, after FinalizeDecoding): surviving
// tokens renumbered 0..n-1 in token order, surviving links as arcs (src, dst, ilabel, olabel, graph, acoustic - cost_offset) in
// link order, final costs of the last frame's tokens.  counts[u] = {n_states, n_arcs, n_final, start state, overflow}.
// Three launches of LAT_P workgroups per utterance (a filter + compaction over ~10^5 links per utterance: with one workgroup
// per utterance and one LDS atomic per surviving arc it took 4 ms for 32 utterances and numbered states and arcs in arrival
// order): count per slice -> new token ids -> arcs.  Slice sums live in the cluster search's work list (rebuilt every frame,
// free between launches); positions come from prefix scans, so the numbering is deterministic.
constexpr int LAT_NT = 256;
constexpr int LAT_P = 32;          // slices per utterance (2 * LAT_P ints of scratch)

struct LatCtx { Lay l; int F, n_tok, l_begin, l_end, p, t0, t1, k0, k1; int* part; };
__device__ __forceinline__ LatCtx lat_ctx(char* state, size_t state_bytes, int max_frames, int max_tok, int max_link, int hash) {
  LatCtx c;
  layout(state + (size_t)blockIdx.y * state_bytes, max_frames, max_tok, max_link, hash, &c.l);
  c.F = c.l.h->n_frames;
  c.n_tok = min(c.l.h->n_tok, max_tok);
  c.l_begin = c.l.link_off[0];
  c.l_end = min(c.l.link_off[2 * c.F + 1], max_link);
  c.p = blockIdx.x;
  // slices start on multiples of 16 elements (16-byte loads of the 1-byte link flags, of 4 token words)
  auto cut = [](int lo, int hi, int q) { return q >= LAT_P ? hi : min(hi, max(lo, (int)(((long long)(hi - lo) * q / LAT_P + lo) & ~15LL))); };
  c.t0 = cut(0, c.n_tok, c.p); c.t1 = cut(0, c.n_tok, c.p + 1);
  c.k0 = cut(c.l_begin, max(c.l_begin, c.l_end), c.p); c.k1 = cut(c.l_begin, max(c.l_begin, c.l_end), c.p + 1);
  c.part = c.l.wlg;
  return c;
}
// 16 consecutive links starting at li0 (li0 % 16 == 0 except at a slice's ragged ends): bit i = link li0 + i survives
__device__ __forceinline__ unsigned lat_links16(const Lay& l, int li0, int lo, int hi) {
  const unsigned INF_BITS = 0x7f800000u;
  unsigned m = 0;
  if (li0 >= hi) return 0;
  if (li0 >= lo && li0 + 16 <= hi && (li0 & 15) == 0) {
    const uint4 v = *reinterpret_cast<const uint4*>(l.link_alive + li0);
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 16; ++i) m |= ((w[i >> 2] >> (8 * (i & 3))) & 0xffu) ? (1u << i) : 0u;
  } else {
    for (int i = 0; i < 16; ++i) { const int li = li0 + i; if (li >= lo && li < hi && l.link_alive[li]) m |= 1u << i; }
  }
  for (unsigned r = m; r; r &= r - 1) {               // the flag says alive: both endpoints must have survived too
    const int i = __ffs(r) - 1, li = li0 + i;
    if (l.tok_extra[l.link_src[li]] == INF_BITS || l.tok_extra[l.link_dst[li]] == INF_BITS) m &= ~(1u << i);
  }
  return m;
}
// 4 consecutive tokens starting at t0q: bit i = token survives
__device__ __forceinline__ unsigned lat_toks4(const Lay& l, int t0q, int lo, int hi) {
  const unsigned INF_BITS = 0x7f800000u;
  unsigned m = 0;
  if (t0q >= hi) return 0;
  if (t0q >= lo && t0q + 4 <= hi && (t0q & 3) == 0) {
    const uint4 v = *reinterpret_cast<const uint4*>(l.tok_extra + t0q);
    m = (v.x != INF_BITS) | ((v.y != INF_BITS) << 1) | ((v.z != INF_BITS) << 2) | ((v.w != INF_BITS) << 3);
  } else {
    for (int i = 0; i < 4; ++i) { const int t = t0q + i; if (t >= lo && t < hi && l.tok_extra[t] != INF_BITS) m |= 1u << i; }
  }
  return m;
}
// exclusive position of this thread's count among the workgroup's counts + the total (LAT_NT threads)
__device__ __forceinline__ int lat_scan(int n, int& total, int* wsum /* [2][LAT_NT / 64] */, int& flip) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int v = n;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) { const int x = __shfl_up(v, off, 64); if (lane >= off) v += x; }
  if (lane == 63) wsum[flip * (LAT_NT / 64) + w] = v;
  __syncthreads();
  int before = 0, tot = 0;
#pragma unroll
  for (int ww = 0; ww < LAT_NT / 64; ++ww) { const int x = wsum[flip * (LAT_NT / 64) + ww]; if (ww < w) before += x; tot += x; }
  total = tot;
  flip ^= 1;
  return before + v - n;
}

__global__ __launch_bounds__(LAT_NT) void wfst_lattice_count_kernel(char* state, size_t state_bytes, int max_frames, int max_tok,
                                                                   int max_link, int hash) {
  __shared__ int acc[2];
  const LatCtx c = lat_ctx(state, state_bytes, max_frames, max_tok, max_link, hash);
  if (threadIdx.x < 2) acc[threadIdx.x] = 0;
  __syncthreads();
  int nt = 0, nk = 0;
  for (int t = c.t0 + 4 * threadIdx.x; t < c.t1; t += 4 * LAT_NT) nt += __popc(lat_toks4(c.l, t, c.t0, c.t1));
  for (int li = c.k0 + 16 * threadIdx.x; li < c.k1; li += 16 * LAT_NT) nk += __popc(lat_links16(c.l, li, c.k0, c.k1));
  for (int off = 32; off; off >>= 1) { nt += __shfl_down(nt, off, 64); nk += __shfl_down(nk, off, 64); }
  if ((threadIdx.x & 63) == 0) { atomicAdd(&acc[0], nt); atomicAdd(&acc[1], nk); }
  __syncthreads();
  if (threadIdx.x < 2) c.part[2 * c.p + threadIdx.x] = acc[threadIdx.x];
}

__global__ __launch_bounds__(LAT_NT) void wfst_lattice_ids_kernel(char* state, size_t state_bytes, int max_frames, int max_tok,
                                                                 int max_link, int hash, int* counts) {
  __shared__ int wsum[2 * (LAT_NT / 64)];
  const LatCtx c = lat_ctx(state, state_bytes, max_frames, max_tok, max_link, hash);
  const unsigned INF_BITS = 0x7f800000u;
  int base = 0, all = 0;
  for (int q = 0; q < LAT_P; ++q) { const int v = c.part[2 * q]; if (q < c.p) base += v; all += v; }
  int* newid = reinterpret_cast<int*>(c.l.tok_prev);     // free after finalize
  int flip = 0;
  for (int tb = c.t0; tb < c.t1; tb += 4 * LAT_NT) {
    const int t = tb + 4 * threadIdx.x;
    const unsigned m = lat_toks4(c.l, t, c.t0, c.t1);
    int tot;
    int k = base + lat_scan(__popc(m), tot, wsum, flip);
    base += tot;
    for (int i = 0; i < 4; ++i) if (t + i >= c.t0 && t + i < c.t1) newid[t + i] = (m >> i) & 1u ? k++ : -1;
  }
  if (c.p == 0 && threadIdx.x == 0) {
    counts[5 * blockIdx.y] = all;
    counts[5 * blockIdx.y + 3] = (c.n_tok > 0 && c.l.tok_extra[0] != INF_BITS) ? 0 : -1;   // token 0 is the start token
  }
}

__global__ __launch_bounds__(LAT_NT) void wfst_lattice_arcs_kernel(Graph g, char* state, size_t state_bytes, int max_frames,
                                                                  int max_tok, int max_link, int hash, int cap_arcs, int cap_final,
                                                                  int* counts, int* a_src, int* a_dst, int* a_il, int* a_ol,
                                                                  float* a_graph, float* a_ac, int* f_state, float* f_cost) {
  __shared__ int wsum[2 * (LAT_NT / 64)];
  const LatCtx c = lat_ctx(state, state_bytes, max_frames, max_tok, max_link, hash);
  const Lay l = c.l;        // a copy, not a reference: see prune_frame
  const int u = blockIdx.y;
  int base = 0, all = 0;
  for (int q = 0; q < LAT_P; ++q) { const int v = c.part[2 * q + 1]; if (q < c.p) base += v; all += v; }
  const int* newid = reinterpret_cast<const int*>(l.tok_prev);
  const size_t ao = (size_t)u * cap_arcs, fo = (size_t)u * cap_final;
  const int nseg = 2 * c.F + 2;                       // link_off[j] <= li < link_off[j + 1]: j odd = emitting links of frame j / 2
  int flip = 0;
  for (int kb = c.k0; kb < c.k1; kb += 16 * LAT_NT) {
    const int li0 = kb + 16 * threadIdx.x;
    const unsigned m = lat_links16(l, li0, c.k0, c.k1);
    int tot;
    int k = base + lat_scan(__popc(m), tot, wsum, flip);
    base += tot;
    for (unsigned r = m; r; r &= r - 1, ++k) {
      if (k >= cap_arcs) break;
      const int li = li0 + __ffs(r) - 1;
      int lo = 0, hi = nseg - 1;                      // last j with link_off[j] <= li
      while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (l.link_off[mid] <= li) lo = mid; else hi = mid - 1; }
      const int a = l.link_arc[li];
      a_src[ao + k] = newid[l.link_src[li]]; a_dst[ao + k] = newid[l.link_dst[li]];
      a_il[ao + k] = g_il(g, a); a_ol[ao + k] = g_ol(g, a);
      a_graph[ao + k] = l.link_graph[li];
      a_ac[ao + k] = (lo & 1) ? l.link_ac[li] - l.cost_offset[lo >> 1] : l.link_ac[li];   // emitting links carry the frame's cost offset
    }
  }
  if (c.p != 0) return;
  // finals: the last frame's surviving tokens with a finite final cost, in token order
  const int t0 = l.tok_off[c.F], t1 = min(l.tok_off[c.F + 1], c.n_tok);
  int nf = 0;
  for (int tb = t0; tb < t1; tb += LAT_NT) {
    const int t = tb + threadIdx.x;
    float fc = INFINITY;
    if (t < t1 && newid[t] >= 0) fc = l.h->has_final ? g.final_cost[l.tok_state[t]] : 0.f;
    const int ok = fc != INFINITY;
    int tot;
    const int k = nf + lat_scan(ok, tot, wsum, flip);
    nf += tot;
    if (ok && k < cap_final) { f_state[fo + k] = newid[t]; f_cost[fo + k] = fc; }
  }
  if (threadIdx.x == 0) {
    counts[5 * u + 1] = min(all, cap_arcs); counts[5 * u + 2] = min(nf, cap_final);
    counts[5 * u + 4] = (all > cap_arcs || nf > cap_final) ? 1 : 0;
  }
}

}  // namespace b2t

using namespace b2t;

extern "C" size_t b2t_wfst_state_bytes(int max_frames, int max_tokens, int max_links, int hash_size) {
  return layout(nullptr, max_frames, max_tokens, max_links, hash_size, nullptr);
}

namespace {
int check_args(const b2t_wfst_graph_t* g, const b2t_wfst_opts_t* o, void* state, int U, const char* what) {
  B2T_REQUIRE(g && o && state && U > 0, "%s: null argument", what);
  B2T_REQUIRE(g->row && g->next && g->n_eps && g->final_cost && g->n_states > 0 &&
              (g->compact ? (g->labels && g->weight_f16) : (g->ilabel && g->olabel && g->weight)), "%s: incomplete graph", what);
  B2T_REQUIRE(o->hash_size >= 64 && (o->hash_size & (o->hash_size - 1)) == 0, "%s: hash_size must be a power of two >= 64", what);
  B2T_REQUIRE(o->max_frames > 0 && o->max_tokens > 0 && o->max_links > 0, "%s: bad capacities", what);
  B2T_REQUIRE(o->beam > 0.f && o->lattice_beam > 0.f && o->max_active > 1 && o->min_active >= 0 && o->min_active <= o->max_active,
              "%s: bad search options", what);
  return 0;
}
Graph to_graph(const b2t_wfst_graph_t* g) {
  return Graph{g->row, g->ilabel, g->olabel, g->weight, g->next, g->n_eps, g->final_cost, g->start,
               g->labels, reinterpret_cast<const _Float16*>(g->weight_f16), g->compact};
}
Opts to_opts(const b2t_wfst_opts_t* o) {
  return Opts{o->beam, o->lattice_beam, o->beam_delta, o->acoustic_scale, o->length_penalty, o->blank_skip_thresh, o->max_active, o->min_active};
}
size_t lds_hash_bytes(const b2t_wfst_opts_t* o) { return o->hash_size <= 16384 ? (size_t)o->hash_size * 2 * sizeof(int) : 0; }   // <= 128 KB of the CU's 160 KB
template <typename K> void allow_lds(K kernel, size_t bytes) {
  if (bytes > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}
}  // namespace

extern "C" int b2t_wfst_reset(const b2t_wfst_graph_t* g, const b2t_wfst_opts_t* o, void* state, int U, void* stream) {
  { int rc = check_args(g, o, state, U, "wfst_reset"); if (rc) return rc; }
  const size_t sb = b2t_wfst_state_bytes(o->max_frames, o->max_tokens, o->max_links, o->hash_size), lds = lds_hash_bytes(o);
  allow_lds(wfst_reset_kernel, lds);
  hipLaunchKernelGGL(wfst_reset_kernel, dim3(U), dim3(NT), lds, as_stream(stream), to_graph(g), (char*)state, sb, to_opts(o),
                     o->max_frames, o->max_tokens, o->max_links, o->hash_size, lds ? 1 : 0);
  B2T_CHECK_LAUNCH("b2t_wfst_reset");
  return 0;
}

// Workgroups per utterance of the search: the largest of 8 / 4 / 2 / 1 that keeps every cluster resident (one 1024-thread
// workgroup per CU, 256 CUs); B2T_WFST_CLUSTER overrides (1 = the single-workgroup kernel with its LDS hash).
// The clusters assume what the hardware does in this partition mode: workgroup b of a launch runs on XCD b % 8.  Probed once
// per process (256 one-wave workgroups report their XCC_ID): if blocks with equal b % 8 do not share an XCD, or the eight
// classes are not on eight different XCDs, the search falls back to one workgroup per utterance.  (The kernel checks again,
// per launch, and refuses to decode on a mismatch.)
__global__ void wfst_xcd_probe_kernel(unsigned* out) { if (threadIdx.x == 0) out[blockIdx.x] = xcc_of(); }
static bool wfst_xcd_roundrobin_ok() {
  static int ok = -1;
  if (ok >= 0) return ok == 1;
  ok = 0;
  unsigned* d = nullptr;
  unsigned h[256];
  if (hipMalloc(reinterpret_cast<void**>(&d), sizeof(h)) != hipSuccess) { (void)hipGetLastError(); return false; }
  hipLaunchKernelGGL(wfst_xcd_probe_kernel, dim3(256), dim3(64), 0, nullptr, d);
  const bool copied = hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess;
  (void)hipFree(d);
  if (!copied) { (void)hipGetLastError(); return false; }
  unsigned seen = 0;
  for (int i = 0; i < 256; ++i) if (h[i] > 7u || h[i] != h[i & 7]) return false;
  for (int i = 0; i < 8; ++i) seen |= 1u << h[i];
  if (seen != 0xffu) return false;
  ok = 1;
  return true;
}

static int g_cluster_override = 0;
// 0 = automatic; 2 .. 32 = that many workgroups per utterance; 1 = the single-workgroup kernel (frame hash in LDS: round 2's
// search, kept as the reference the cluster search is tested against); -1 = the cluster kernel with ONE member
extern "C" int b2t_wfst_set_cluster(int G) { g_cluster_override = G < -1 ? 0 : G; return 0; }
static int wfst_forced_cluster() {
  static const int env = getenv("B2T_WFST_CLUSTER") ? atoi(getenv("B2T_WFST_CLUSTER")) : 0;
  return g_cluster_override ? g_cluster_override : env;
}
extern "C" int b2t_wfst_cluster_size(int U) {
  const int forced = wfst_forced_cluster();
  if (forced == -1) return 1;
  if (forced >= 1) return forced >= 32 ? 32 : forced >= 16 ? 16 : forced >= 8 ? 8 : forced >= 4 ? 4 : forced >= 2 ? 2 : 1;
  // 8 workgroups per utterance where they are all resident.  Larger clusters work (b2t_wfst_set_cluster(16 / 32): up to a whole
  // XCD per utterance, tested against the single-workgroup search) but buy nothing: one utterance takes 11.7 / 11.1 / 11.7 ms
  // with 8 / 16 / 32 workgroups, eight take 14.9 / 14.1 / 14.9 -- beyond 8 members a frame is its ~6 cluster barriers and the
  // chains of dependent L2 round trips between them (~100 us), not the walk over its tokens and arcs.
  // one workgroup (1024 threads) per CU: the members of every cluster must be co-resident (their barrier is an L2 spin counter),
  // so the slot count is the device's CU count, not a constant (a partition or a smaller part has fewer)
  static int slots = -1;
  if (slots < 0) {
    int dev = 0; hipDeviceProp_t p;
    slots = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) ? p.multiProcessorCount : 256;
  }
  for (int G = 8; G > 1; G >>= 1) if ((U + 7) / 8 * 8 * G <= slots) return wfst_xcd_roundrobin_ok() ? G : 1;
  return 1;
}

extern "C" int b2t_wfst_search_f32(const b2t_wfst_graph_t* g, const b2t_wfst_opts_t* o, void* state, const float* logp,
                                   const int32_t* lens, int U, int T, int C, void* stream) {
  { int rc = check_args(g, o, state, U, "wfst_search"); if (rc) return rc; }
  B2T_REQUIRE(logp && T > 0 && C > 1 && C <= MAX_C, "wfst_search: bad logp shape T=%d C=%d", T, C);
  const size_t sb = b2t_wfst_state_bytes(o->max_frames, o->max_tokens, o->max_links, o->hash_size), lds = lds_hash_bytes(o);
  const int G = b2t_wfst_cluster_size(U);
  // More utterances than clusters fit: still the cluster kernel, with one member each (its barriers then cost an atomic, and
  // its one-pass epsilon closure and single claim-relax-link walk make it faster than the single-workgroup kernel with its
  // ~45 __syncthreads per frame: 55.4 against 63.1 ms for 256 utterances).
  if (G > 1 || wfst_forced_cluster() != 1) {
    const int grid = (U + 7) / 8 * 8 * G;
    static const bool no_stamp = getenv("B2T_WFST_STAMPED") && atoi(getenv("B2T_WFST_STAMPED")) == 0;   // A/B knob: clear a hash per frame (round 3)
    const int stamped = (!no_stamp && g->n_states < (1 << 27)) ? 1 : 0;
#define B2T_CLUSTER_GO(ST_, CP_)                                                                                        \
    hipLaunchKernelGGL((wfst_cluster_kernel<ST_, CP_>), dim3(grid), dim3(NT), 0, as_stream(stream), to_graph(g), (char*)state, sb, to_opts(o), \
                       o->max_frames, o->max_tokens, o->max_links, o->hash_size, G, U, logp, lens, T, C, stamped)
    if (stamped) { if (g->compact) B2T_CLUSTER_GO(true, true); else B2T_CLUSTER_GO(true, false); }
    else { if (g->compact) B2T_CLUSTER_GO(false, true); else B2T_CLUSTER_GO(false, false); }
#undef B2T_CLUSTER_GO
    B2T_CHECK_LAUNCH("b2t_wfst_search_f32 (cluster)");
    return 0;
  }
  allow_lds(wfst_search_kernel, lds);
  hipLaunchKernelGGL(wfst_search_kernel, dim3(U), dim3(NT), lds, as_stream(stream), to_graph(g), (char*)state, sb, to_opts(o),
                     o->max_frames, o->max_tokens, o->max_links, o->hash_size, lds ? 1 : 0, logp, lens, T, C);
  B2T_CHECK_LAUNCH("b2t_wfst_search_f32");
  return 0;
}

extern "C" int b2t_wfst_best_path(const b2t_wfst_graph_t* g, const b2t_wfst_opts_t* o, const void* state, int U, int use_final,
                                  int max_len, int32_t* alignment, int32_t* align_frame, int32_t* n_align, int32_t* words,
                                  int32_t* n_words, float* costs, void* stream) {
  { int rc = check_args(g, o, const_cast<void*>(state), U, "wfst_best_path"); if (rc) return rc; }
  B2T_REQUIRE(max_len > 0 && alignment && align_frame && n_align && words && n_words && costs, "wfst_best_path: null output");
  const size_t sb = b2t_wfst_state_bytes(o->max_frames, o->max_tokens, o->max_links, o->hash_size);
  hipLaunchKernelGGL(wfst_best_path_kernel, dim3(U), dim3(BP_NT), 0, as_stream(stream), to_graph(g), (char*)const_cast<void*>(state), sb,
                     o->max_frames, o->max_tokens, o->max_links, o->hash_size, use_final, max_len, alignment, align_frame, n_align,
                     words, n_words, costs);
  B2T_CHECK_LAUNCH("b2t_wfst_best_path");
  return 0;
}

extern "C" int b2t_wfst_finalize(const b2t_wfst_graph_t* g, const b2t_wfst_opts_t* o, void* state, int U, void* stream) {
  { int rc = check_args(g, o, state, U, "wfst_finalize"); if (rc) return rc; }
  const size_t sb = b2t_wfst_state_bytes(o->max_frames, o->max_tokens, o->max_links, o->hash_size);
  {   // the utterance's cluster finalizes where the cluster searched (B2T_WFST_FIN_CLUSTER=0, read per call: one workgroup per utterance)
    const char* e = getenv("B2T_WFST_FIN_CLUSTER");
    const int G = (e && atoi(e) == 0) ? 1 : b2t_wfst_cluster_size(U);
    // (cluster kernels spin on L2 barriers: the launch must be fully resident.  b2t_wfst_cluster_size sizes clusters for ONE cluster
    //  kernel on the device at a time -- search, prune and finalize launches of decode streams that run concurrently must be
    //  serialised by the caller (WfstSearch does: one stream per searcher, passes behind the search) or take B2T_WFST_*_CLUSTER=0;
    //  a partly resident launch ends in CBAR_SPIN_LIMIT with overflow | 32, never in a hang)
    B2T_REQUIRE(G <= 32, "wfst_finalize: clusters of at most 32 workgroups (scratch layout), got %d", G);
    if (G > 1) {
      const int grid = (U + 7) / 8 * 8 * G;
      hipLaunchKernelGGL(wfst_finalize_cluster_kernel, dim3(grid), dim3(NT), 0, as_stream(stream), to_graph(g), (char*)state, sb, to_opts(o),
                         o->max_frames, o->max_tokens, o->max_links, o->hash_size, G, U);
      B2T_CHECK_LAUNCH("b2t_wfst_finalize (cluster)");
      return 0;
    }
  }
  allow_lds(wfst_finalize_kernel, PRUNE_LDS_WORDS * sizeof(unsigned));
  hipLaunchKernelGGL(wfst_finalize_kernel, dim3(U), dim3(NT), PRUNE_LDS_WORDS * sizeof(unsigned), as_stream(stream), to_graph(g), (char*)state,
                     sb, to_opts(o), o->max_frames, o->max_tokens, o->max_links, o->hash_size);
  B2T_CHECK_LAUNCH("b2t_wfst_finalize");
  return 0;
}

extern "C" int b2t_wfst_prune(const b2t_wfst_graph_t* g, const b2t_wfst_opts_t* o, void* state, int U, float delta, float min_fill,
                              void* stream) {
  { int rc = check_args(g, o, state, U, "wfst_prune"); if (rc) return rc; }
  B2T_REQUIRE(delta >= 0.f && min_fill >= 0.f && min_fill <= 1.f, "wfst_prune: bad delta / min_fill");
  const size_t sb = b2t_wfst_state_bytes(o->max_frames, o->max_tokens, o->max_links, o->hash_size);
  {   // the utterance's cluster prunes where the cluster searches (B2T_WFST_PRUNE_CLUSTER=0, read per call: one workgroup per utterance)
    const char* e = getenv("B2T_WFST_PRUNE_CLUSTER");
    const int G = (e && atoi(e) == 0) ? 1 : b2t_wfst_cluster_size(U);
    const size_t lds = (size_t)3 * (o->max_frames + 3) * sizeof(int);
    B2T_REQUIRE(G <= 32, "wfst_prune: clusters of at most 32 workgroups (tots[80] / scratch layout), got %d", G);
    static_assert(WLG_CAP >= 64 + 32 * 8 + 32, "the cluster compaction's per-member scratch lives in the work list");
    if (G > 1 && lds <= 96 * 1024) {
      const int grid = (U + 7) / 8 * 8 * G;
      allow_lds(wfst_prune_cluster_kernel, lds);
      hipLaunchKernelGGL(wfst_prune_cluster_kernel, dim3(grid), dim3(NT), lds, as_stream(stream), to_graph(g), (char*)state, sb, to_opts(o),
                         o->max_frames, o->max_tokens, o->max_links, o->hash_size, delta, min_fill, G, U);
      B2T_CHECK_LAUNCH("b2t_wfst_prune (cluster)");
      return 0;
    }
  }
  allow_lds(wfst_prune_kernel, PRUNE_LDS_WORDS * sizeof(unsigned));
  hipLaunchKernelGGL(wfst_prune_kernel, dim3(U), dim3(NT), PRUNE_LDS_WORDS * sizeof(unsigned), as_stream(stream), to_graph(g), (char*)state, sb,
                     to_opts(o), o->max_frames, o->max_tokens, o->max_links, o->hash_size, delta, min_fill);
  B2T_CHECK_LAUNCH("b2t_wfst_prune");
  return 0;
}

extern "C" int b2t_wfst_lattice(const b2t_wfst_graph_t* g, const b2t_wfst_opts_t* o, void* state, int U, int cap_arcs, int cap_final,
                                int32_t* counts, int32_t* src, int32_t* dst, int32_t* ilabel, int32_t* olabel, float* graph,
                                float* acoustic, int32_t* final_state, float* final_cost, void* stream) {
  { int rc = check_args(g, o, state, U, "wfst_lattice"); if (rc) return rc; }
  B2T_REQUIRE(cap_arcs > 0 && cap_final > 0 && counts && src && dst && ilabel && olabel && graph && acoustic && final_state && final_cost,
              "wfst_lattice: null output / zero capacity");
  const size_t sb = b2t_wfst_state_bytes(o->max_frames, o->max_tokens, o->max_links, o->hash_size);
  static_assert(2 * LAT_P <= WLG_CAP, "slice sums live in the work list");
  hipLaunchKernelGGL(wfst_lattice_count_kernel, dim3(LAT_P, U), dim3(LAT_NT), 0, as_stream(stream), (char*)state, sb, o->max_frames,
                     o->max_tokens, o->max_links, o->hash_size);
  hipLaunchKernelGGL(wfst_lattice_ids_kernel, dim3(LAT_P, U), dim3(LAT_NT), 0, as_stream(stream), (char*)state, sb, o->max_frames,
                     o->max_tokens, o->max_links, o->hash_size, counts);
  hipLaunchKernelGGL(wfst_lattice_arcs_kernel, dim3(LAT_P, U), dim3(LAT_NT), 0, as_stream(stream), to_graph(g), (char*)state, sb,
                     o->max_frames, o->max_tokens, o->max_links, o->hash_size, cap_arcs, cap_final, counts, src, dst, ilabel, olabel,
                     graph, acoustic, final_state, final_cost);
  B2T_CHECK_LAUNCH("b2t_wfst_lattice");
  return 0;
}

// Host views into one utterance's state block (offsets in bytes from the block's start), for copying the lattice out.
extern "C" int b2t_wfst_state_offsets(int max_frames, int max_tokens, int max_links, int hash_size, long long* off16) {
  B2T_REQUIRE(off16 != nullptr, "wfst_state_offsets: null output");
  Lay l;
  char* base = reinterpret_cast<char*>(0x1000);   // any non-null base: only differences are used
  layout(base, max_frames, max_tokens, max_links, hash_size, &l);
  const char* ptrs[16] = {(char*)l.h, (char*)l.mapping, (char*)l.tok_off, (char*)l.link_off, (char*)l.cost_offset, (char*)l.tok_state,
                          (char*)l.tok_cost, (char*)l.tok_extra, (char*)l.link_src, (char*)l.link_dst, (char*)l.link_arc,
                          (char*)l.link_ac, (char*)l.link_graph, (char*)l.link_alive, (char*)l.tok_best, (char*)l.last_prob};
  for (int i = 0; i < 16; ++i) off16[i] = (long long)(ptrs[i] - base);
  return 0;
}
