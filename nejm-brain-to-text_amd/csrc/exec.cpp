// exec.cpp — the model pass (GRUDecoder.forward, model_training/rnn_model.py:88-134, and its backward, rnn_trainer.py:547)
// issued from C++: every GEMM / sweep / reduction launch and every stream-event edge of the pipelined execution plan
// comes from here, so that one pass is ONE call across the C ABI instead of ~250 ctypes calls.
//
// Execution plan.  A GRU layer is a strictly serial chain over time and one layer's persistent sweep keeps only
// (H/16) x ceil(B/16) workgroups busy, each mostly waiting on the inter-workgroup hand-off.  The time axis is cut into
// chunks and the layers are software-pipelined over them on per-layer HIP streams: while layer l sweeps chunk c, layer
// l+1 runs its input-projection GEMM + sweep on chunk c-1; in the backward pass the weight-gradient GEMMs of layer l
// run on that layer's GEMM stream while the layers below are still sweeping.  Dependencies are HIP events; the
// caller's stream joins the side streams before the function returns (nothing synchronises with the host).
#include <vector>
#include <algorithm>
#include "common.h"

namespace b2t {
namespace {

constexpr int MAXL = B2T_MAX_LAYERS;
constexpr int MAXC = 32;   // time chunks

struct ProfRec { int kind; double flops; hipEvent_t e0, e1; };

}  // namespace
}  // namespace b2t

struct b2t_exec {
  int L = 0;
  hipStream_t s_sweep[b2t::MAXL] = {}, s_gemm[b2t::MAXL] = {};
  std::vector<hipEvent_t> pool;   // ordering events (timing disabled), handed out round-robin within a pass
  size_t next_ev = 0;
  bool profile = false;
  std::vector<b2t::ProfRec> recs;
  std::vector<hipEvent_t> tpool;  // timing events
  size_t next_tev = 0;
};

namespace b2t {
namespace {

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

int out_T(const b2t_model_t* m, int T) { return m->patch > 0 ? (T - m->patch) / m->stride + 1 : T; }
int in0(const b2t_model_t* m) { return m->patch > 0 ? m->F * m->patch : m->F; }

// Number of K slices so that a weight-gradient GEMM (few 128x128 output tiles, very long K) fills the chip: ~4
// workgroups per CU on 256 CUs, each slice at least 256 deep.
int splitk_for(int M, int N, long long K, int target_blocks = 1024) {
  const int tiles = ((M + 127) / 128) * ((N + 127) / 128);
  long long v = std::min<long long>(target_blocks / std::max(1, tiles), K / 256);
  return (int)std::max<long long>(1, v);
}

// Time chunks of the layer pipeline: `chunks` equal parts (at least 16 steps each).
int make_chunks(int Tp, int chunks, int (*out)[2]) {
  int n = std::max(1, std::min(std::min(chunks, MAXC), Tp / 16));
  const int ch = (Tp + n - 1) / n;
  int k = 0;
  for (int t0 = 0; t0 < Tp; t0 += ch) { out[k][0] = t0; out[k][1] = std::min(Tp, t0 + ch); ++k; }
  return k;
}

// ---- workspace layout (deterministic in (model, pass): forward and backward carve the same addresses) --------------
struct Layout {
  float *U, *Ud, *out[MAXL], *outd[MAXL], *gi[MAXL], *res[MAXL], *slab_gi[MAXL];
  float *dY[MAXL], *dG[MAXL], *dh_init, *carry[MAXL], *scratch[MAXL], *whh_t[MAXL], *dU, *dV, *day_slab, *day_bslab;
  float *slab[MAXL], *slab_head, *s4[MAXL], *cs_layer[MAXL], *cs_head, *cs_day, *cs_h0;
  size_t bytes;
};

size_t colsum_ws_floats(long long rows, int cols) { return b2t_colsum_ws_bytes(rows, cols) / sizeof(float); }

void carve(const b2t_model_t* m, const b2t_pass_t* p, char* base, Layout& w) {
  const size_t B = p->B, T = p->T, F = m->F, H = m->H, L = m->L, C = m->C;
  const size_t Tp = out_T(m, p->T), In0 = in0(m);
  size_t off = 0;
  auto take = [&](size_t nfloats) {
    float* r = reinterpret_cast<float*>(base + off);
    off += align_up(nfloats * sizeof(float), 256);
    return r;
  };
  w.U = take(B * T * F);
  w.Ud = p->in_drop > 0.f ? take(B * T * F) : w.U;
  for (size_t l = 0; l < L; ++l) {
    w.out[l] = take((Tp + 1) * B * H);
    w.outd[l] = (p->rnn_drop > 0.f && l + 1 < L) ? take((Tp + 1) * B * H) : w.out[l];
    w.gi[l] = take(Tp * B * 3 * H);
  }
  // split-K slabs of the streaming-sized input projections (chunks of n*B <= 512 rows, see b2t_model_forward)
  for (size_t l = 0; l < L; ++l) {
    const size_t sk = l == 0 ? (In0 >= 2048 ? std::max<size_t>(1, std::min<size_t>(16, In0 / 448)) : 0)
                             : (H >= 384 ? std::max<size_t>(1, H / 192) : 0);
    w.slab_gi[l] = sk > 1 ? take(sk * std::min<size_t>(512, Tp * B) * 3 * H) : nullptr;
  }
  if (!p->save) { w.bytes = off; return; }
  for (size_t l = 0; l < L; ++l) w.res[l] = take(Tp * B * 4 * H);
  const size_t K = Tp * B;
  for (size_t l = 0; l < L; ++l) {
    w.dY[l] = take(Tp * B * H);
    w.dG[l] = take(Tp * B * 4 * H);
    w.carry[l] = take(2 * B * H);
    w.scratch[l] = take(B * H);
    w.whh_t[l] = take(H * 3 * H);
    const size_t In = l == 0 ? In0 : H;
    const size_t a = (size_t)splitk_for(3 * H, H, K) * 3 * H * H, b = (size_t)splitk_for(3 * H, In, K) * 3 * H * In;
    const size_t b2 = (size_t)splitk_for(2 * H, In, K) * 2 * H * In;   // the two-GEMM form of dW_ih (odd H)
    w.slab[l] = take(std::max(a, std::max(b, b2)));
    w.s4[l] = take(4 * H);
    w.cs_layer[l] = take(colsum_ws_floats(K, 4 * H) + 4);
  }
  w.dh_init = take(L * B * H);
  w.dU = take(B * T * F);
  w.dV = m->patch > 0 ? take(B * Tp * In0) : nullptr;
  w.day_slab = take(B * F * F);
  w.day_bslab = take(B * align_up(F, 4));
  w.slab_head = take((size_t)splitk_for(C, H, K) * C * H);
  w.cs_head = take(colsum_ws_floats(B * Tp, C) + 4);
  w.cs_day = take(B * colsum_ws_floats(T, F) + 4);
  w.cs_h0 = take(colsum_ws_floats(L * B, H) + 4);
  w.bytes = off;
}

// ---- pass context: streams, events, profiling ----------------------------------------------------------------------
struct Ctx {
  b2t_exec* ex;
  hipStream_t main;
  bool bf16_gemm;
  int rc = 0;

  hipEvent_t record(hipStream_t s) {
    if (ex->next_ev == ex->pool.size()) {
      hipEvent_t e;
      if (check_hip(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate")) { rc = 1; return nullptr; }
      ex->pool.push_back(e);
    }
    hipEvent_t e = ex->pool[ex->next_ev++];
    if (check_hip(hipEventRecord(e, s), "hipEventRecord")) rc = 1;
    return e;
  }
  void wait(hipStream_t s, hipEvent_t e) {
    if (e && check_hip(hipStreamWaitEvent(s, e, 0), "hipStreamWaitEvent")) rc = 1;
  }
  hipEvent_t tev() {
    if (ex->next_tev == ex->tpool.size()) {
      hipEvent_t e;
      if (check_hip(hipEventCreate(&e), "hipEventCreate")) { rc = 1; return nullptr; }
      ex->tpool.push_back(e);
    }
    return ex->tpool[ex->next_tev++];
  }
  struct Scope {
    Ctx& c; hipStream_t s; int kind; double flops; hipEvent_t e0 = nullptr;
    Scope(Ctx& c_, hipStream_t s_, int kind_, double flops_) : c(c_), s(s_), kind(kind_), flops(flops_) {
      if (c.ex->profile) { e0 = c.tev(); if (e0) (void)hipEventRecord(e0, s); }
    }
    ~Scope() {
      if (c.ex->profile && e0) {
        hipEvent_t e1 = c.tev();
        if (e1) { (void)hipEventRecord(e1, s); c.ex->recs.push_back(ProfRec{kind, flops, e0, e1}); }
      }
    }
  };

  // C = A.B^T through b2t_gemm_f32 / b2t_gemm_bf16_f32.  splitk > 1: partial products go to `slab` ([splitk][M*N]) and
  // are summed deterministically into C by b2t_slab_reduce_f32 (weight gradients, K = T*B; streaming-sized projections).
  void gemm(hipStream_t s, b2t_gemm_desc d, int splitk = 1, float* slab = nullptr, int accumulate = 0) {
    if (rc) return;
    void* st = reinterpret_cast<void*>(s);
    const int kind = (bf16_gemm ? 4 : 0) + (d.a_kcontig ? 2 : 0) + (d.b_kcontig ? 1 : 0);
    const double flops = 2.0 * d.M * d.N * (double)d.K * (d.Z > 0 ? d.Z : 1);
    if (d.Z <= 0) d.Z = 1;
    if (splitk > 1) {
      float* Cdst = d.C;
      if (d.Z != 1 || d.epilogue != 0 || d.c_div != 0 || d.c_s0 != d.N) {
        set_error("exec: split-K gemm supports Z=1, dense row-major C, no epilogue"); rc = 2; return;
      }
      d.C = slab; d.splitk = splitk; d.c_ks = (long long)d.M * d.N; d.accumulate = 0;
      {
        Scope sc(*this, s, kind, flops);
        rc = bf16_gemm ? b2t_gemm_bf16_f32(&d, st) : b2t_gemm_f32(&d, st);
      }
      if (!rc) rc = b2t_slab_reduce_f32(slab, splitk, (long long)d.M * d.N, Cdst, accumulate, st);
      return;
    }
    d.accumulate = accumulate;
    Scope sc(*this, s, kind, flops);
    rc = bf16_gemm ? b2t_gemm_bf16_f32(&d, st) : b2t_gemm_f32(&d, st);
  }
  void call(int r) { if (!rc) rc = r; }
};

b2t_gemm_desc gd(const float* A, const float* Bm, float* C, int M, int N, int K) {
  b2t_gemm_desc d;
  memset(&d, 0, sizeof(d));
  d.A = A; d.B = Bm; d.C = C; d.M = M; d.N = N; d.K = K; d.Z = 1;
  d.a_kcontig = 1; d.b_kcontig = 1;
  return d;
}

uint64_t mix_seed(uint64_t seed, uint64_t k) { return seed * 1000003ull + k; }

int check_common(const b2t_exec* ex, const b2t_model_t* m, const b2t_pass_t* p, const char* what) {
  B2T_REQUIRE(ex && m && p, "%s: null argument", what);
  B2T_REQUIRE(m->L >= 1 && m->L <= MAXL && m->L <= ex->L, "%s: %d layers (executor has %d, max %d)", what, m->L, ex->L, MAXL);
  B2T_REQUIRE(m->H > 0 && m->H % 16 == 0 && m->F > 0 && m->F % 4 == 0 && m->C > 0, "%s: bad dims F=%d H=%d C=%d", what, m->F, m->H, m->C);
  B2T_REQUIRE(p->B > 0 && p->T > 0 && out_T(m, p->T) > 0, "%s: bad batch B=%d T=%d (sequence shorter than patch_size?)", what, p->B, p->T);
  B2T_REQUIRE(m->patch == 0 || m->stride > 0, "%s: patch_size %d needs patch_stride > 0", what, m->patch);
  return 0;
}

}  // namespace
}  // namespace b2t

using namespace b2t;

extern "C" int b2t_exec_create(int n_layers, b2t_exec** out) {
  B2T_REQUIRE(out && n_layers >= 1 && n_layers <= MAXL, "exec_create: 1..%d layers", MAXL);
  b2t_exec* ex = new b2t_exec();
  ex->L = n_layers;
  for (int l = 0; l < n_layers; ++l) {
    if (check_hip(hipStreamCreateWithFlags(&ex->s_sweep[l], hipStreamNonBlocking), "hipStreamCreate") ||
        check_hip(hipStreamCreateWithFlags(&ex->s_gemm[l], hipStreamNonBlocking), "hipStreamCreate")) {
      delete ex;
      return 1;
    }
  }
  *out = ex;
  return 0;
}

extern "C" int b2t_exec_destroy(b2t_exec* ex) {
  if (!ex) return 0;
  for (int l = 0; l < ex->L; ++l) {
    if (ex->s_sweep[l]) (void)hipStreamDestroy(ex->s_sweep[l]);
    if (ex->s_gemm[l]) (void)hipStreamDestroy(ex->s_gemm[l]);
  }
  for (hipEvent_t e : ex->pool) (void)hipEventDestroy(e);
  for (hipEvent_t e : ex->tpool) (void)hipEventDestroy(e);
  delete ex;
  return 0;
}

extern "C" size_t b2t_exec_sync_bytes(int n_layers) { return (size_t)2 * n_layers * b2t_gru_sync_bytes(0); }

extern "C" size_t b2t_pass_ws_bytes(const b2t_model_t* m, const b2t_pass_t* p) {
  if (!m || !p || p->B <= 0 || p->T <= 0 || out_T(m, p->T) <= 0 || m->L < 1 || m->L > MAXL) return 0;
  Layout w;
  carve(m, p, nullptr, w);
  return w.bytes + 256;
}

extern "C" int b2t_exec_profile(b2t_exec* ex, int on) {
  B2T_REQUIRE(ex, "exec_profile: null executor");
  ex->profile = on != 0;
  return 0;
}

extern "C" int b2t_exec_profile_read(b2t_exec* ex, int* kind_host, double* flops_host, float* ms_host, int cap) {
  if (!ex) { set_error("exec_profile_read: null executor"); return -1; }
  if (check_hip(hipDeviceSynchronize(), "exec_profile_read: sync")) return -1;
  int n = 0;
  for (const ProfRec& r : ex->recs) {
    if (n >= cap) break;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) ms = 0.f;
    kind_host[n] = r.kind; flops_host[n] = r.flops; ms_host[n] = ms; ++n;
  }
  ex->recs.clear();
  ex->next_tev = 0;
  return n;
}

// ------------------------------------------------------------------------------------------------------------------
// forward: day layer -> (patch) -> L x GRU -> head
// ------------------------------------------------------------------------------------------------------------------
extern "C" int b2t_model_forward(b2t_exec* ex, const b2t_model_t* prm, const b2t_pass_t* p, const float* x,
                                 const int32_t* day_idx, const float* states, float* logits, float* hidden, void* ws,
                                 void* sync_ws, void* stream) {
  { int rc = check_common(ex, prm, p, "model_forward"); if (rc) return rc; }
  B2T_REQUIRE(x && day_idx && logits && hidden && ws, "model_forward: null buffer");
  const int B = p->B, T = p->T, F = prm->F, H = prm->H, L = prm->L, Cc = prm->C;
  const int Tp = out_T(prm, T), In0 = in0(prm);
  const int mode = p->fwd_mode;
  B2T_REQUIRE((mode & 0xff) == 0 || sync_ws, "model_forward: sync_ws is required for the persistent sweeps");
  Layout w;
  carve(prm, p, reinterpret_cast<char*>(ws), w);
  Ctx c{ex, as_stream(stream), p->bf16_gemm != 0};
  ex->next_ev = 0;
  hipStream_t main = c.main;
  void* mainp = stream;
  const size_t sync_block = b2t_gru_sync_bytes(0);
  auto sync_of = [&](int l) { return sync_ws ? reinterpret_cast<char*>(sync_ws) + (size_t)l * sync_block : nullptr; };

  // 1. day layer: U[b] = softsign(x[b] @ W[day[b]] + c[day[b]])   (rnn_model.py:95-99); the [B,512,512] gather of the
  //    reference does not exist: the GEMM indexes the day weights by day_idx (b_zmap)
  {
    b2t_gemm_desc d = gd(x, prm->day_w, w.U, T, F, F);
    d.Z = B; d.a_s0 = F; d.a_sz = (long long)T * F; d.b_kcontig = 0; d.b_s0 = F; d.b_sz = prm->day_w_stride;
    d.c_s0 = F; d.c_sz = (long long)T * F; d.bias = prm->day_b; d.bias_sz = prm->day_b_stride; d.b_zmap = day_idx; d.epilogue = 1;
    c.gemm(main, d);
  }
  if (p->in_drop > 0.f)
    c.call(b2t_dropout_f32(w.U, w.Ud, (long long)B * T * F, p->in_drop, mix_seed(p->seed, 17), 0, mainp));

  int chunks[MAXC][2];
  const int nc = make_chunks(Tp, p->chunks, chunks);
  const bool piped = nc > 1;
  const long long a_s0_l0 = prm->patch > 0 ? (long long)prm->stride * F : F;

  if (piped) {
    hipEvent_t ev0 = c.record(main);
    for (int l = 0; l < L; ++l) { c.wait(ex->s_sweep[l], ev0); c.wait(ex->s_gemm[l], ev0); }
  }
  // slot 0 of out[l] = initial state, so out[l][0:T'] is the h_{t-1} matrix (on the layer's sweep stream)
  for (int l = 0; l < L; ++l) {
    hipStream_t s = piped ? ex->s_sweep[l] : main;
    if (states) c.call(check_hip(hipMemcpyAsync(w.out[l], states + (size_t)l * B * H, sizeof(float) * B * H, hipMemcpyDeviceToDevice, s), "model_forward: state copy"));
    else c.call(b2t_broadcast_rows_f32(prm->h0, w.out[l], B, H, s));
  }
  hipEvent_t ev_sw[MAXL][MAXC] = {};
  // cells (chunk c, layer l) are enqueued diagonal by diagonal (c + l), a topological order in which no stream waits on
  // work that is queued behind it
  for (int diag = 0; diag < nc + L - 1 && !c.rc; ++diag) {
    for (int l = 0; l < L; ++l) {
      const int ci = diag - l;
      if (ci < 0 || ci >= nc) continue;
      const int t0 = chunks[ci][0], t1 = chunks[ci][1], n = t1 - t0;
      // (bounding the forward sweeps in flight to 4 / 3 by sharing sweep streams was measured: 23.8 / 24.1 ms against 23.2)
      hipStream_t sg = piped ? ex->s_gemm[l] : main, ss = piped ? ex->s_sweep[l] : main;
      // 2. input projection gi = in_t W_ih^T + b_ih for this chunk, time-major [T'][B][3H]
      if (l == 0) {
        if ((long long)n * B <= 512 && In0 >= 2048) {
          // streaming-sized calls (a few frames, patch input K = 7168): one GEMM over all (t, b) rows through the
          // two-level row map, K split over the chip (as B per-sentence GEMMs the K loop runs serially in 18 workgroups)
          b2t_gemm_desc d = gd(w.Ud + (long long)t0 * a_s0_l0, prm->w_ih[0], w.gi[0] + (long long)t0 * B * 3 * H, n * B, 3 * H, In0);
          d.a_div = B; d.a_s1 = a_s0_l0; d.a_s0 = (long long)T * F; d.b_s0 = In0; d.c_s0 = 3 * H; d.bias = prm->b_ih[0];
          c.gemm(sg, d, std::max(1, std::min(16, In0 / 448)), w.slab_gi[0]);
        } else {
          b2t_gemm_desc d = gd(w.Ud + (long long)t0 * a_s0_l0, prm->w_ih[0], w.gi[0] + (long long)t0 * B * 3 * H, n, 3 * H, In0);
          d.Z = B; d.a_s0 = a_s0_l0; d.a_sz = (long long)T * F; d.b_s0 = In0; d.c_s0 = (long long)B * 3 * H; d.c_sz = 3 * H;
          d.bias = prm->b_ih[0];
          c.gemm(sg, d);
        }
      } else {
        if (piped) c.wait(sg, ev_sw[l - 1][ci]);
        const float* src = w.out[l - 1];
        if (w.outd[l - 1] != w.out[l - 1]) {   // nn.GRU inter-layer dropout (rnn_model.py:70)
          c.call(b2t_dropout_f32(w.out[l - 1] + (long long)(1 + t0) * B * H, w.outd[l - 1] + (long long)(1 + t0) * B * H,
                                 (long long)n * B * H, p->rnn_drop, mix_seed(p->seed, 101 + (l - 1)), (long long)t0 * B * H,
                                 reinterpret_cast<void*>(sg)));
          src = w.outd[l - 1];
        }
        b2t_gemm_desc d = gd(src + (long long)(1 + t0) * B * H, prm->w_ih[l], w.gi[l] + (long long)t0 * B * 3 * H, n * B, 3 * H, H);
        d.a_s0 = H; d.b_s0 = H; d.c_s0 = 3 * H; d.bias = prm->b_ih[l];
        const bool small = (long long)n * B <= 512 && H >= 384;   // streaming-sized call: split K (one 128-row tile otherwise)
        if (small) c.gemm(sg, d, std::max(1, H / 192), w.slab_gi[l]);
        else c.gemm(sg, d);
      }
      hipEvent_t ev_gi = piped ? c.record(sg) : nullptr;
      // 3. recurrent sweep over the chunk, continuing from out[l][t0] = h_{t0-1}
      if (piped) c.wait(ss, ev_gi);
      if (!c.rc) {
        Ctx::Scope sc(c, ss, 8, 2.0 * n * B * 3.0 * H * H);
        c.call(b2t_gru_layer_fwd_f32(w.gi[l] + (long long)t0 * B * 3 * H, prm->w_hh[l], prm->b_hh[l], w.out[l] + (long long)t0 * B * H,
                                     w.out[l] + (long long)(1 + t0) * B * H, p->save ? w.res[l] + (long long)t0 * B * 4 * H : nullptr,
                                     t1 == Tp ? hidden + (size_t)l * B * H : nullptr, n, B, H, mode, sync_of(l),
                                     reinterpret_cast<void*>(ss)));
      }
      if (piped) ev_sw[l][ci] = c.record(ss);
    }
  }
  // One join is enough: the last chunk of the top layer's sweep transitively depends on every GEMM and sweep enqueued
  // above (each wait is a barrier packet the command processor works through one by one, ~50 us apiece).
  if (piped) c.wait(main, ev_sw[L - 1][nc - 1]);

  // 4. head: logits[b,t,:] = out W^T + b  (rnn_model.py:129), written batch-first
  {
    b2t_gemm_desc d = gd(w.out[L - 1] + (long long)B * H, prm->out_w, logits, Tp * B, Cc, H);
    d.a_s0 = H; d.b_s0 = H; d.c_div = B; d.c_s1 = Cc; d.c_s0 = (long long)Tp * Cc; d.bias = prm->out_b;
    c.gemm(main, d);
  }
  return c.rc;
}

// ------------------------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------------------------
namespace b2t {
namespace {

// dW_hh = dGh^T h_prev, dW_ih = dGi^T in, bias gradients = column sums of dG (layer l) over the time rows [t0, t1):
// the whole sequence at once (t0 = 0, t1 = T', accumulate = 0), or chunk by chunk as soon as a chunk is swept (the first
// chunk processed overwrites, later ones accumulate in a fixed order: deterministic); `final` copies the bias sums out.
void layer_weight_grads(Ctx& c, hipStream_t s, const b2t_model_t* prm, const b2t_model_t* grd, const b2t_pass_t* p,
                        Layout& w, int l, int t0, int t1, int accumulate, bool final) {
  const int B = p->B, T = p->T, F = prm->F, H = prm->H;
  const long long K = (long long)(t1 - t0) * B;
  const long long a0 = (long long)t0 * B * 4 * H;
  void* sp = reinterpret_cast<void*>(s);
  {
    b2t_gemm_desc d = gd(w.dG[l] + a0, w.out[l] + (long long)t0 * B * H, grd->w_hh[l], 3 * H, H, (int)K);
    d.a_kcontig = 0; d.a_s0 = 4 * H; d.b_kcontig = 0; d.b_s0 = H; d.c_s0 = H;
    c.gemm(s, d, splitk_for(3 * H, H, K), w.slab[l], accumulate);
  }
  int In; const float* inp; long long b_s0, b_s1 = 0; int b_div = 0;
  if (l == 0) {
    In = in0(prm);
    b_div = B; b_s1 = prm->patch > 0 ? (long long)prm->stride * F : F; b_s0 = (long long)T * F;
    inp = w.Ud + (long long)t0 * b_s1;
  } else {
    In = H; inp = w.outd[l - 1] + (long long)(1 + t0) * B * H;   // skip the initial-state slot
    b_s0 = H;
  }
  auto wih = [&](int M, long long a_off, long long c_off, int brk, int gap) {
    b2t_gemm_desc d = gd(w.dG[l] + a0 + a_off, inp, grd->w_ih[l] + c_off, M, In, (int)K);
    d.a_kcontig = 0; d.a_s0 = 4 * H; d.b_kcontig = 0; d.b_s0 = b_s0; d.b_s1 = b_s1; d.b_div = b_div; d.c_s0 = In;
    d.a_brk = brk; d.a_gap = gap;
    c.gemm(s, d, splitk_for(M, In, K), w.slab[l], accumulate);
  };
  if ((2 * H) % 128 == 0 && (3 * H) % 128 == 0) {   // dGi^T as ONE operand with a gap along m
    wih(3 * H, 0, 0, 2 * H, H);
  } else {
    wih(2 * H, 0, 0, 0, 0);
    wih(H, 3 * H, (long long)2 * H * In, 0, 0);
  }
  c.call(b2t_colsum_f32(w.dG[l] + a0, K, 4 * H, 4 * H, w.s4[l], accumulate, w.cs_layer[l], 1, 0, 0, sp));   // (s_r, s_z, s_nr, s_n)
  if (!final) return;
  auto cp = [&](float* dst, const float* src, size_t n) {
    c.call(check_hip(hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, s), "model_backward: bias gradient copy"));
  };
  cp(grd->b_ih[l], w.s4[l], 2 * H);
  cp(grd->b_ih[l] + 2 * H, w.s4[l] + 3 * H, H);
  cp(grd->b_hh[l], w.s4[l], 3 * H);
}

}  // namespace
}  // namespace b2t

extern "C" int b2t_model_backward(b2t_exec* ex, const b2t_model_t* prm, const b2t_model_t* grd, const b2t_pass_t* p,
                                  const float* x, const int32_t* day_idx, const float* dlogits, int ldd,
                                  const float* dhidden, float* dstates, int custom_states, void* ws, void* sync_ws,
                                  b2t_bucket_cb bucket_cb, void* user, void* stream) {
  { int rc = check_common(ex, prm, p, "model_backward"); if (rc) return rc; }
  B2T_REQUIRE(grd && x && day_idx && dlogits && ws && p->save, "model_backward: null buffer / forward ran without save");
  B2T_REQUIRE(ldd >= prm->C && ldd % 4 == 0, "model_backward: ldd=%d must be a multiple of 4 and >= C", ldd);
  const int B = p->B, T = p->T, F = prm->F, H = prm->H, L = prm->L, Cc = prm->C;
  const int Tp = out_T(prm, T), In0 = in0(prm);
  const long long M = (long long)Tp * B;
  const int mode = p->bwd_mode;
  B2T_REQUIRE((mode & 0xff) == 0 || sync_ws, "model_backward: sync_ws is required for the persistent sweeps");
  Layout w;
  carve(prm, p, reinterpret_cast<char*>(ws), w);
  Ctx c{ex, as_stream(stream), p->bf16_gemm != 0};
  ex->next_ev = 0;
  hipStream_t main = c.main;
  void* mainp = stream;
  const size_t sync_block = b2t_gru_sync_bytes(0);
  auto sync_of = [&](int l) { return sync_ws ? reinterpret_cast<char*>(sync_ws) + (size_t)(L + l) * sync_block : nullptr; };
  auto cb = [&](int id, hipStream_t s) { if (bucket_cb && !c.rc) bucket_cb(user, id, reinterpret_cast<void*>(s)); };

  int chunks[MAXC][2];
  const int nc = make_chunks(Tp, p->chunks_bwd > 0 ? p->chunks_bwd : p->chunks, chunks);
  // (One chunk -- shapes whose sweeps cannot be co-resident, e.g. H = 768 -- runs everything on the caller's stream.  Putting
  // the weight-gradient GEMMs of layer l on a side stream under the sweep of layer l - 1 was measured: C3 fp32 19.7 -> 23.0 ms,
  // bf16 operands 12.7 -> 15.1 ms: a 768-unit sweep workgroup needs a CU's whole register file, and GEMM workgroups that
  // arrive first keep its row group from becoming resident.)
  const bool piped = nc > 1;

  // head: d_out[t,b,:] = dlogits[b,t,:] W_out ; dW_out = dlogits^T out ; db_out = colsum
  {
    b2t_gemm_desc d = gd(dlogits, prm->out_w, w.dY[L - 1], (int)M, H, Cc);
    d.a_div = B; d.a_s1 = ldd; d.a_s0 = (long long)Tp * ldd; d.b_kcontig = 0; d.b_s0 = H; d.c_s0 = H;
    c.gemm(main, d);
  }
  hipEvent_t ev_top = piped ? c.record(main) : nullptr;
  {
    b2t_gemm_desc d = gd(dlogits, w.out[L - 1] + (long long)B * H, grd->out_w, Cc, H, (int)M);
    d.a_kcontig = 0; d.a_div = B; d.a_s1 = ldd; d.a_s0 = (long long)Tp * ldd; d.b_kcontig = 0; d.b_s0 = H; d.c_s0 = H;
    c.gemm(main, d, splitk_for(Cc, H, M), w.slab_head);
  }
  c.call(b2t_colsum_f32(dlogits, M, Cc, ldd, grd->out_b, 0, w.cs_head, 1, 0, 0, mainp));
  cb(0, main);

  hipEvent_t ev_wt[MAXL] = {};
  if (piped) {
    // W_hh^T for the backward sweeps depends on the parameters only: enqueued BEFORE the streams wait for the head (it
    // runs while the CTC kernel has the chip to itself instead of in front of the first backward sweep)
    for (int l = 0; l < L; ++l) {
      c.call(b2t_transpose_f32(prm->w_hh[l], w.whh_t[l], 3 * H, H, reinterpret_cast<void*>(ex->s_gemm[l])));
      ev_wt[l] = c.record(ex->s_gemm[l]);
    }
    for (int l = 0; l < L; ++l) { c.wait(ex->s_sweep[l], ev_top); c.wait(ex->s_gemm[l], ev_top); }
  }

  // dIn = dGi W_ih for rows of chunk [t0, t0+n): into dY[l-1] (l > 0) or dU / dV (l == 0).
  // Day-layer backward chunk by chunk (no patching, no input dropout): the Softsign backward rides in the epilogue of
  // layer 0's dX GEMM, the per-sample day-gradient GEMM and bias sums accumulate chunk after chunk behind it, so that
  // only the last chunk's share is left behind the last backward sweep (it was 1.4 ms of tail as whole-sequence passes).
  const bool fast_day = prm->patch == 0 && !(p->in_drop > 0.f);
  const long long bias_ld = (long long)align_up(F, 4);
  auto dx_gemm = [&](hipStream_t s, int l, int t0, int n) {
    // dGi = dG[:, 0:2H] ++ dG[:, 3H:4H] is ONE A operand with a gap at k = 2H (H % 16 == 0 makes 2H a multiple of the k tile)
    const int N = l > 0 ? H : In0;
    b2t_gemm_desc d = gd(w.dG[l] + (long long)t0 * B * 4 * H, prm->w_ih[l], nullptr, n * B, N, 3 * H);
    d.a_s0 = 4 * H; d.b_kcontig = 0; d.b_s0 = N; d.a_brk = 2 * H; d.a_gap = H;
    if (l > 0) { d.C = w.dY[l - 1] + (long long)t0 * B * H; d.c_s0 = H; }
    else if (prm->patch > 0) { d.C = w.dV + (long long)t0 * In0; d.c_div = B; d.c_s1 = In0; d.c_s0 = (long long)Tp * In0; }
    else {
      d.C = w.dU + (long long)t0 * F; d.c_div = B; d.c_s1 = F; d.c_s0 = (long long)T * F;
      if (fast_day) { d.epilogue = 2; d.ep_aux = w.U + (long long)t0 * F; }   // dpre = dU * (1 - |U|)^2
    }
    c.gemm(s, d);
    if (l == 0 && fast_day) {
      const int first = t0 + n == Tp ? 0 : 1;   // the top time chunk is swept first: it overwrites, the others accumulate
      b2t_gemm_desc d = gd(x + (long long)t0 * F, w.dU + (long long)t0 * F, w.day_slab, F, F, n);
      d.Z = B; d.a_kcontig = 0; d.a_s0 = F; d.a_sz = (long long)T * F; d.b_kcontig = 0; d.b_s0 = F; d.b_sz = (long long)T * F;
      d.c_s0 = F; d.c_sz = (long long)F * F;
      c.gemm(s, d, 1, nullptr, first);
      c.call(b2t_colsum_f32(w.dU + (long long)t0 * F, n, F, F, w.day_bslab, first, w.cs_day, B, (long long)T * F, bias_ld,
                            reinterpret_cast<void*>(s)));
    }
  };

  hipEvent_t ev_dx[MAXL][MAXC] = {}, ev_bs[MAXL][MAXC] = {};
  for (int diag = 0; diag < nc + L - 1 && !c.rc; ++diag) {
    for (int l = L - 1; l >= 0; --l) {
      const int ci = nc - 1 - (diag - (L - 1 - l));
      if (ci < 0 || ci >= nc) continue;
      const int t0 = chunks[ci][0], t1 = chunks[ci][1], n = t1 - t0;
      hipStream_t ss = piped ? ex->s_sweep[l] : main, sg = piped ? ex->s_gemm[l] : main;
      void* ssp = reinterpret_cast<void*>(ss);
      if (piped) {
        if (l < L - 1) c.wait(ss, ev_dx[l + 1][ci]);
        if (ci == nc - 1) c.wait(ss, ev_wt[l]);
      } else if (ci == nc - 1) {
        c.call(b2t_transpose_f32(prm->w_hh[l], w.whh_t[l], 3 * H, H, ssp));
      }
      if (p->rnn_drop > 0.f && l < L - 1)   // gradient through the inter-layer dropout mask
        c.call(b2t_dropout_f32(w.dY[l] + (long long)t0 * B * H, w.dY[l] + (long long)t0 * B * H, (long long)n * B * H, p->rnn_drop,
                               mix_seed(p->seed, 101 + l), (long long)t0 * B * H, ssp));
      const float* dh_last = ci == nc - 1 ? (dhidden ? dhidden + (size_t)l * B * H : nullptr) : w.carry[l] + (size_t)((ci + 1) % 2) * B * H;
      float* dh_out = ci == 0 ? w.dh_init + (size_t)l * B * H : w.carry[l] + (size_t)(ci % 2) * B * H;
      if (!c.rc) {
        Ctx::Scope sc(c, ss, 9, 2.0 * n * B * 3.0 * H * H);
        c.call(b2t_gru_layer_bwd_f32(w.dY[l] + (long long)t0 * B * H, dh_last, w.res[l] + (long long)t0 * B * 4 * H,
                                     w.out[l] + (long long)(1 + t0) * B * H, w.out[l] + (long long)t0 * B * H, w.whh_t[l],
                                     w.dG[l] + (long long)t0 * B * 4 * H, dh_out, w.scratch[l], n, B, H, mode, sync_of(l), ssp));
      }
      if (piped) ev_bs[l][ci] = c.record(ss);
      if (piped) c.wait(sg, ev_bs[l][ci]);
      dx_gemm(sg, l, t0, n);
      if (piped) ev_dx[l][ci] = c.record(sg);
      // Weight gradients on a GEMM stream.  Layers >= 1: the whole layer once its last chunk is swept (they overlap the
      // sweeps of the layers below; per-chunk launches there were measured slower: more launches competing for the
      // sweeps' CUs).  Layer 0 has no layer below to hide behind -- its 100 GFLOP used to sit in the step's tail -- so its
      // weight gradients accumulate chunk by chunk on the top layer's GEMM stream (idle by then), and only the last
      // chunk's share follows the last sweep.
      const bool per_chunk = piped && ((p->wgrad_chunk_mask >> l) & 1);
      if (per_chunk || ci == 0) {
        hipStream_t swg = !piped ? main : ((l == 0 && L > 1) ? ex->s_gemm[L - 1] : ex->s_gemm[l]);
        if (piped) c.wait(swg, ev_bs[l][ci]);
        if (per_chunk) layer_weight_grads(c, swg, prm, grd, p, w, l, t0, t1, ci == nc - 1 ? 0 : 1, ci == 0);
        else layer_weight_grads(c, swg, prm, grd, p, w, l, 0, Tp, 0, true);
        if (ci == 0) cb(1 + l, swg);
      }
    }
  }

  // layer-0 input gradient -> day layer (on layer 0's GEMM stream: its dU/dV GEMMs are already ordered there)
  {
    hipStream_t s = piped ? ex->s_gemm[0] : main;
    void* sp = reinterpret_cast<void*>(s);
    if (!fast_day) {
      if (prm->patch > 0) c.call(b2t_patch_fold_f32(w.dV, w.dU, B, T, F, Tp, prm->patch, prm->stride, sp));
      if (p->in_drop > 0.f) c.call(b2t_dropout_f32(w.dU, w.dU, (long long)B * T * F, p->in_drop, mix_seed(p->seed, 17), 0, sp));
      c.call(b2t_softsign_bwd_f32(w.U, w.dU, (long long)B * T * F, sp));   // dpre = dU * (1-|U|)^2, in place
      // per-sample partial day gradients, then deterministic reduction by day
      b2t_gemm_desc d = gd(x, w.dU, w.day_slab, F, F, T);
      d.Z = B; d.a_kcontig = 0; d.a_s0 = F; d.a_sz = (long long)T * F; d.b_kcontig = 0; d.b_s0 = F; d.b_sz = (long long)T * F;
      d.c_s0 = F; d.c_sz = (long long)F * F;
      c.gemm(s, d);
      c.call(b2t_colsum_f32(w.dU, T, F, F, w.day_bslab, 0, w.cs_day, B, (long long)T * F, bias_ld, sp));
    }
    c.call(b2t_day_reduce_f32(w.day_slab, day_idx, B, (long long)F * F, grd->day_w, grd->day_w_stride, sp));
    c.call(b2t_day_reduce_f32(w.day_bslab, day_idx, B, bias_ld, grd->day_b, grd->day_b_stride, sp));
    cb(L + 2, s);
  }
  // h0 gradient: sum over layers and batch rows of the carry after t=0 (rnn_model.py:86,123)
  auto h0_grad = [&](hipStream_t s) {
    void* sp = reinterpret_cast<void*>(s);
    if (!custom_states) c.call(b2t_colsum_f32(w.dh_init, (long long)L * B, H, H, grd->h0, 0, w.cs_h0, 1, 0, 0, sp));
    else c.call(check_hip(hipMemsetAsync(grd->h0, 0, sizeof(float) * H, s), "model_backward: h0 gradient"));
    if (dstates) c.call(check_hip(hipMemcpyAsync(dstates, w.dh_init, sizeof(float) * L * B * H, hipMemcpyDeviceToDevice, s), "model_backward: dstates"));
    cb(L + 1, s);
  };
  if (piped && L > 2) {
    // Every sweep stream's last launch is followed by a GEMM on that layer's GEMM stream, and the weight-gradient
    // streams are GEMM streams: joining the L GEMM streams joins everything.  The streams that finish early (layers
    // 1 .. L-2) are joined into one of them while the last two are still busy; the caller's stream then waits for three
    // events instead of L.  The h0 reduction rides on that idle stream (layer 0's sweep is the last one to finish).
    hipStream_t s1 = ex->s_gemm[1];
    for (int l = 2; l < L - 1; ++l) c.wait(s1, c.record(ex->s_gemm[l]));
    c.wait(s1, ev_bs[0][0]);
    h0_grad(s1);
    c.wait(main, c.record(s1));
    c.wait(main, c.record(ex->s_gemm[L - 1]));
    c.wait(main, c.record(ex->s_gemm[0]));
  } else {
    if (piped) for (int l = 0; l < L; ++l) c.wait(main, c.record(ex->s_gemm[l]));
    h0_grad(main);
  }
  return c.rc;
}
