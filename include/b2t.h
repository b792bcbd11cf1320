/* b2t.h — C ABI of libb2t_hip.so, the MI355X (gfx950) hot path of the brain-to-text decoder.
 *
 * Drop-in boundary: the Python host (nejm-brain-to-text_amd/{rnn_model,rnn_trainer,
 * data_augmentations,lm_decoder}.py) keeps the reference's call surfaces and calls these
 * entry points through ctypes with raw device pointers and the caller's HIP stream.
 * Plain C: no C++ types, no exceptions across the boundary, no torch types.
 *
 * Conventions
 *   - return 0 = OK, non-zero = error; message via b2t_last_error() (thread-local).
 *   - every pointer is a DEVICE pointer unless the name ends in _host.
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *     all calls are asynchronous on it; the library never synchronises the device, never
 *     allocates or frees device memory on the training path and keeps no pointers after return.
 *   - all floating point is fp32; index arrays are int32.
 *   - internal activations of the GRU stack are TIME-MAJOR [T][B][*]; model inputs/outputs
 *     (features, logits) are batch-first [B][T][*] like the reference's tensors.
 *
 * Each entry point cites the reference call site it replaces (paths relative to the
 * reference repository root).
 */
#ifndef B2T_H
#define B2T_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define B2T_VERSION 2

int b2t_version(void);
const char* b2t_last_error(void);

/* ---- a1/a2: augmentation + Gaussian smoothing ------------------------------------------
 * Replaces BrainToTextDecoder_Trainer.transform_data (model_training/rnn_trainer.py:436-484)
 * and gauss_smooth (model_training/data_augmentations.py:6-37) with ONE fused pass.
 *   x [B][T][F]  ->  y [B][T_out][F]
 *   y[b,t,f] = sum_j taps[j] * xn[b, t + j - left, f]   (zero outside [0, T-cut))
 *   xn[b,t',f] = x[b,t'+cut,f] + white_std*N(b,t'+cut,f) + offset_std*N'(b,f)
 * padding_mode 0 = 'same' (T_out = T-cut, left=(ntaps-1)/2), 1 = 'valid' (T_out = T-cut-ntaps+1,
 * left = 0).  Noise is counter-based Philox4x32-10 keyed by (seed, element index); with
 * white_std = offset_std = 0 no noise is generated (validation / evaluation path).
 * white_noise / offset_noise: optional device tensors [B][T][F] / [B][F] of pre-drawn N(0,1)
 * (used by parity tests to inject the reference's draws); when non-NULL they replace Philox.
 * taps_host: ntaps <= 33 floats on the HOST (copied into the kernel argument). */
int b2t_augment_smooth_f32(const float* x, float* y, int B, int T, int F, int cut,
                           float white_std, float offset_std, uint64_t seed,
                           const float* white_noise, const float* offset_noise,
                           const float* taps_host, int ntaps, int padding_mode, void* stream);

/* ---- generic fp32 MFMA GEMM (v_mfma_f32_32x32x2_f32, exact fp32) ------------------------
 * C[z][m][n] (+)= sum_k A(z,m,k) * B(z,n,k) (+ bias[n]) with an optional fused epilogue.
 * Used for: day layer (rnn_model.py:98-99), GRU input projections and their gradients
 * (rnn_model.py:126 -> nn.GRU), output head (rnn_model.py:129).
 * Addressing (elements):  A(m,k): a_kcontig ? A[z*a_sz + rowoff_a(m) + k] : A[z*a_sz + rowoff_a(k) + m]
 * where rowoff_x(i) = (i / x_div) * x_s1 + (i % x_div) * x_s0  (x_div <= 0: i * x_s0).
 * Same for B with n.  C row m: rowoff_c(m), columns contiguous.
 * b_zmap (optional, int32[Z]): B batch index = b_zmap[z] (day-indexed weights, no gather copy:
 * replaces the torch.stack at rnn_model.py:95).
 * epilogue: 0 = none, 1 = softsign(v) (rnn_model.py:47,99), 2 = v * (1 - |u|)^2 with u = ep_aux at C's index (the
 * Softsign backward fused into the GEMM that produces the gradient wrt its output).  accumulate: C += result. */
typedef struct b2t_gemm_desc {
  const float* A; const float* B; float* C; const float* bias;
  int M, N, K, Z;
  int a_kcontig, b_kcontig;
  long long a_s0, a_s1; int a_div; long long a_sz;
  long long b_s0, b_s1; int b_div; long long b_sz;
  long long c_s0, c_s1; int c_div; long long c_sz;
  const int* b_zmap; long long bias_sz;   /* bias row of batch z = bias + b_zmap[z]*bias_sz (0: shared) */
  int epilogue; int accumulate;
  /* split-K (weight gradients: few output tiles, K = T*B): splitk > 1 launches Z*splitk slices; slice ks
   * reduces k in [ks*kc, min(K,(ks+1)*kc)) (kc = ceil(K/splitk) rounded up to 16) into the slab
   * C + z*c_sz + ks*c_ks; the caller sums the slabs deterministically (b2t_colsum_f32).
   * splitk <= 1: one pass.  With M <= 64 rows b2t_gemm_f32 then streams the weights once through a kernel whose 8 waves
   * split K (a streamed frame of a few dozen utterances: 5-8x faster than a 128-row tile); splitk < 0 asks for the tile
   * kernel's k order regardless (passes that must agree bit for bit with differently chunked passes of the same data). */
  int splitk; long long c_ks;
  /* gap in A's contiguous index i (k if a_kcontig, else m): for i >= a_brk the element is read at i + a_gap.
   * Lets dGi = dG[:, 0:2H] ++ dG[:, 3H:4H] be ONE operand (a_brk = 2H, a_gap = H).  a_brk = 0: off; must be a
   * multiple of 16 (k) / 128 (m, with M % 128 == 0). */
  int a_brk; int a_gap;
  const float* ep_aux;   /* epilogue 2 only: same layout as C (z*c_sz + row offset + column) */
  /* b2t_gemm_f32 with an m-contiguous A (a_kcontig = 0) only, NULL otherwise: a_sum[(z*max(1,splitk) + ks)*a_sum_ks + m] =
   * sum over the slice's k of A[k][m] (m = A's logical index, after a_brk), written by the workgroups of the first
   * column tile from the A tiles they stage anyway.  The bias gradients of a GRU layer (column sums of dG, K = T*B rows)
   * come out of the weight-gradient GEMM that reads dG as its A operand instead of a separate pass over dG.
   * b2t_gemm_bf16p_f32 takes it too: its pack pass over A writes one slice per 64 k, a_sum[(k / 64)*a_sum_ks + m]
   * (ceil(K / 64) slices whatever splitk is), summed from the fp32 values before they are rounded. */
  float* a_sum; long long a_sum_ks;
  /* b2t_gemm_f32 with splitk > 1, Z = 1, N % 4 == 0 (NULL otherwise): in-kernel slab reduction.  ks_counters: one zeroed
   * 32-bit word per 128x128 output tile (left zero again); the LAST slice workgroup of a tile to finish sums the tile's
   * slabs in slice order (the order b2t_slab_reduce_f32 uses: bit-identical) into ks_out, a dense row-major [M][N]
   * matrix (+= when ks_accumulate).  No separate reduction launch; C still names the slab area. */
  unsigned* ks_counters; float* ks_out; int ks_accumulate;
} b2t_gemm_desc;
int b2t_gemm_f32(const b2t_gemm_desc* d, void* stream);
/* The same GEMM with the operands rounded to bf16 (nearest-even) on their way to the matrix cores, fp32 accumulation and
 * fp32 output: the matmul regime of the reference's `use_amp` / autocast(bfloat16) training (rnn_trainer.py:535).  Tensors
 * stay fp32 in memory.  Split-K chunks and a_brk along k are multiples of 32 here. */
int b2t_gemm_bf16_f32(const b2t_gemm_desc* d, void* stream);
/* The same product in two passes: both operands are first packed through the descriptor's addressing into dense
 * k-contiguous bf16 matrices in `ws` (b2t_gemm_bf16p_ws_bytes(M, N, K) bytes, 256-byte aligned, caller-owned scratch: no
 * contents survive), then multiplied on 128x128x64 tiles that load bf16 straight into LDS (256x256x64 where those fill the
 * chip): 2.5-3x the one-pass kernel on the training step's shapes.  Same numerics contract (operands rounded to bf16
 * nearest-even, fp32 accumulation, fp32 output and epilogues).  Z-batched descriptors (Z > 1, optional b_zmap / bias_sz: the
 * day layer's per-sentence products, rnn_model.py:95-99) pack every matrix of the batch: b2t_gemm_bf16p_ws_bytes_z bytes,
 * no split-K / a_sum / a_brk. */
size_t b2t_gemm_bf16p_ws_bytes(int M, int N, int K);
size_t b2t_gemm_bf16p_ws_bytes_z(int M, int N, int K, int Z);
int b2t_gemm_bf16p_f32(const b2t_gemm_desc* d, void* ws, size_t ws_bytes, void* stream);

/* ---- elementwise helpers ------------------------------------------------------------------
 * softsign backward (rnn_model.py:99 autograd): du[i] *= (1-|u[i]|)^2, in place. */
int b2t_softsign_bwd_f32(const float* u, float* du, long long n, void* stream);
/* adjusted_lens = ((n_time_steps - patch_size) / patch_stride + 1).to(torch.int32) (rnn_trainer.py:532, :705; fp32 division,
 * truncation), n_time_steps int32 or int64 [B]; patch_size 0: the lengths themselves (the reference divides by zero there). */
int b2t_adjusted_lens_i32(const void* n_time_steps, int is_int64, int B, int patch_size, int patch_stride, int32_t* out, void* stream);
/* column sums: out[z*out_sz + c] (+)= sum_r x[z*x_sz + r*ld + c], r<rows, c<cols (bias gradients);
 * two deterministic stages; ws needs Z * b2t_colsum_ws_bytes(rows, cols) bytes. */
size_t b2t_colsum_ws_bytes(long long rows, int cols);   /* per batch entry z */
int b2t_colsum_f32(const float* x, long long rows, int cols, long long ld, float* out,
                   int accumulate, float* ws, int Z, long long x_sz, long long out_sz, void* stream);
/* split-K slab reduction: out[i] (+)= sum_{s<nslab} slab[s*n + i], slabs added in index order (deterministic). */
int b2t_slab_reduce_f32(const float* slab, int nslab, long long n, float* out, int accumulate, void* stream);
/* per-day reduction of per-sample partial gradients (day layer, rnn_model.py:95-98 autograd):
 * out[d*out_stride + i] = sum_{b: day_idx[b]==d} slab[b*n + i] for every day present in day_idx
 * (summed in batch order: deterministic); days absent from the batch are not touched. */
int b2t_day_reduce_f32(const float* slab, const int32_t* day_idx, int B, long long n, float* out,
                       long long out_stride, void* stream);
/* patch fold (adjoint of the unfold at rnn_model.py:106-119): du[b,t,f] = sum over windows.
 * dv [B][Tp][patch*F] -> du [B][T][F] */
int b2t_patch_fold_f32(const float* dv, float* du, int B, int T, int F, int Tp,
                       int patch, int stride, void* stream);
/* The same fold followed by the backward of the input dropout (the forward's mask: Philox stream of b2t_dropout_f32 over the
 * elements of [B][T][F] from element 0; drop_p = 0: none) and the Softsign backward du *= (1 - |u|)^2 (rnn_model.py:99-103 through
 * autograd), one pass instead of three; every element goes through the same operations in the same order. */
int b2t_patch_fold_day_bwd_f32(const float* dv, const float* u, float* du, int B, int T, int F, int Tp, int patch, int stride,
                               float drop_p, uint64_t seed, void* stream);
/* dropout (rnn_model.py:102-103, nn.GRU inter-layer dropout :70): y = x*mask/(1-p), Philox mask
 * keyed by (seed, elem0 + index) so that a tensor processed in chunks gets the mask of the whole tensor;
 * forward and backward apply the same mask. In place allowed. */
int b2t_dropout_f32(const float* x, float* y, long long n, float p, uint64_t seed, long long elem0,
                    void* stream);
/* random-walk augmentation (rnn_trainer.py:464-465): y += cumsum(w, axis) over tensors viewed as [outer][n][inner], summed
 * in index order along n (torch.cumsum's CPU order). */
int b2t_cumsum_add_f32(const float* w, float* y, long long outer, int n, long long inner, void* stream);
/* the factors themselves: y[i] = 0 or 1/(1-p), the mask b2t_dropout_f32 applies for the same (seed, elem0) */
int b2t_dropout_mask_f32(float* y, long long n, float p, uint64_t seed, long long elem0, void* stream);

/* ---- f1: batch assembly from a device-resident flat dataset (replaces the per-trial HDF5 reads + pad_sequence of
 * model_training/dataset.py:100-159).  flat: rows of W 4-byte elements (all trials' frames / labels back to back);
 * out[b][t][0:W] = flat[row_off[b] + t][0:W] for t < lens[b], zeros up to T_out.  16-byte accesses when W % 4 == 0
 * (flat and out then 16-byte aligned). */
int b2t_batch_gather_b32(const void* flat, const int64_t* row_off, const int32_t* lens, void* out, int B, int T_out,
                         int W, void* stream);

/* ---- a5: GRU layer sweep (torch.nn.GRU at rnn_model.py:65-72,126) --------------------------
 * One layer, all T steps.  gi [T][B][3H] = W_ih x_t + b_ih (precomputed by b2t_gemm_f32),
 * gate order r,z,n.  h_init [B][H].  w_hh [3H][H], b_hh [3H].
 * out [T][B][H]; reserve [T][B][4H] = (r,z,n,gh_n) saved for the backward sweep (may be NULL
 * for inference).  h_last [B][H] (optional) = out[T-1].
 * mode: 0 = step-launch kernels (one launch per time step), 1 = persistent sweep (one launch,
 * W_hh slices resident in registers, agent-scope counter hand-off of h_t between workgroups).
 * mode 1 | B2T_GRU_BF16: the recurrent product takes bf16 operands (h_{t-1} / dG_{t+1} and the W_hh slice rounded to
 * nearest-even bf16, v_mfma_f32_16x16x16_bf16, fp32 accumulate; gates and everything stored stay fp32): the reference's
 * autocast(bfloat16) regime for the GRU, opt-in.
 * sync_ws: device scratch of b2t_gru_sync_bytes(T) bytes (persistent mode), zeroed ONCE by the owner (self-cleaning
 * afterwards); word 0 is a sticky error word (1 = a bounded hand-off spin gave up: results invalid). */
#define B2T_MAX_LAYERS_ 8   /* = B2T_MAX_LAYERS below */
#define B2T_GRU_BF16 0x100
#define B2T_GRU_LOCAL 0x400   /* persistent sweeps, H <= 512, B <= 64: XCD-local hand-off (row group r on XCD (2r + parity) & 7, counters as L2 atomics, tiles written through and read back from that XCD's L2); ignored where the dispatch probe fails */
#define B2T_GRU_PARITY 0x800  /* with B2T_GRU_LOCAL: the layer's parity */
#define B2T_GRU_WIDE 0x200   /* with B2T_GRU_BF16: 32 hidden units per workgroup (half the workgroups per sweep), H % 32 == 0, H <= 512 */
#define B2T_GRU_PAIRED 0x1000 /* b2t_gru_layer_bwd_f32, mode 1, exact fp32, H % 32 == 0, H <= 512, B <= 64 (other shapes: ignored): the backward sweep with its W_hh^T slice in LDS -- one 512-thread workgroup per CU owns 16 dh columns of TWO row groups, ~half the registers per lane, so that a GEMM workgroup stays resident next to it; XCD-local hand-off, pair p on XCD (4 p + set) & 7 */
#define B2T_GRU_WAVE 0x8000   /* b2t_pass_t fwd_mode / bwd_mode, with mode 1 | B2T_GRU_BF16 and bf16_gemm: the L sweeps of the pass as ONE launch, layer l + 1 a step or two behind layer l (b2t_gru_wave_fwd_f32 / _bwd_f32 below); ignored for shapes b2t_gru_wave_supported refuses */
#define B2T_GRU_SET_SHIFT 13  /* with B2T_GRU_PAIRED: bits 13-14 = the XCD set (0..3) of this sweep; at most ONE paired sweep of a set may be in flight */
size_t b2t_gru_sync_bytes(int T);
size_t b2t_gru_ws_bytes(int T, int B, int H);   /* = b2t_gru_sync_bytes (kept for callers that size by shape) */
/* Persistent mode only: copies the sweep's error word to the host and synchronises the stream.
 * *status_host = 0: clean; 1: a bounded hand-off spin gave up (results of that sweep are invalid). */
int b2t_gru_sync_status(const void* sync_ws, int T, int B, int* status_host, void* stream);
int b2t_gru_layer_fwd_f32(const float* gi, const float* w_hh, const float* b_hh,
                          const float* h_init, float* out, float* reserve, float* h_last,
                          int T, int B, int H, int mode, void* sync_ws, void* stream);
/* The same forward sweep of layer l that ALSO produces the next layer's input projection (torch.nn.GRU applies W_ih of
 * layer l+1 to the outputs of layer l, rnn_model.py:65-72,126): gi_next [T][B][3H] = out_t w_ih_next^T + b_ih_next, computed
 * in the sweep's idle matrix-core slots from the h fragments a step already holds in LDS -- the projection GEMM of layers >= 1
 * then does not exist.  Exact fp32, persistent mode only (mode = 1, optionally | B2T_GRU_LOCAL | B2T_GRU_PARITY), H <= 512.
 * w_ih_next [3H][H], b_ih_next [3H].  Not usable when dropout sits between the two layers (training with rnn_dropout > 0). */
int b2t_gru_layer_fwd_fused_f32(const float* gi, const float* w_hh, const float* b_hh, const float* h_init, float* out,
                                float* reserve, float* h_last, const float* w_ih_next, const float* b_ih_next,
                                float* gi_next, int T, int B, int H, int mode, void* sync_ws, void* stream);
/* Backward sweep.  dY [T][B][H] grad wrt this layer's outputs (plus dh_last
 * [B][H] optional grad wrt the final state).  w_hh_t [H][3H] is W_hh transposed (b2t_transpose).
 * dG [T][B][4H] = (dr_pre, dz_pre, dn_pre*r, dn_pre): dGh = cols [0,3H), dGi = cols [0,2H)+[3H,4H).
 * dh_init [B][H] = carry after t=0.  carry_ws: [B][H] floats scratch (mode 0 only). */
int b2t_gru_layer_bwd_f32(const float* dY, const float* dh_last, const float* reserve,
                          const float* out, const float* h_init, const float* w_hh_t,
                          float* dG, float* dh_init, float* carry_ws,
                          int T, int B, int H, int mode, void* sync_ws, void* stream);
/* ---- a5 / a8, round 6: the whole GRU stack's sweeps as ONE launch per direction -- the step-granular layer wavefront
 * (torch.nn.GRU(num_layers = L) at rnn_model.py:65-72,126 under torch.autocast(bfloat16), rnn_trainer.py:527, and its autograd
 * backward).  bf16 operands on the matrix cores (h_{t-1}, dG_{t+1}, W_hh, W_ih of the layers >= 1 rounded to nearest-even bf16),
 * fp32 accumulation, gates, states and everything stored.  All L x H / 16 workgroups are resident at once; layer l + 1 runs one or
 * two time steps behind layer l; the input projections of the layers >= 1 (forward) and the gradients wrt their inputs (backward)
 * are computed inside the sweep of the consuming layer and never exist in memory; nn.GRU's inter-layer dropout (drop_p, Philox
 * stream of b2t_dropout_f32 with seed[l] / elem0: flat element (t, row, unit) of layer l's output) is applied where layer l
 * publishes its state.  H % 16 == 0, H <= 768, B <= 64, L x H / 16 <= CUs of the device (b2t_gru_wave_supported).
 *   forward : gi0 [T][B][3H] (layer 0's projection with b_ih[0]), w_hh / b_hh [l], w_ih / b_ih [l >= 1], h_init[l] [B][H]
 *             -> out[l] [T][B][H], outd[l] = dropout(out[l]) (drop_p > 0, l < L - 1), reserve[l] [T][B][4H] (may be NULL)
 *   backward: dY_top [T][B][H], dh_last [L][B][H] or NULL, w_hh_t[l] = W_hh^T [H][3H], w_ih_t[l >= 1] = W_ih^T [H][3H], h_init / out /
 *             reserve as the forward left them -> dG[l] [T][B][4H] = (dr, dz, dn r, dn), dh_init [L][B][H]
 * ws: b2t_gru_wave_ws_bytes bytes, any contents (hand-off rings + counters, cleared by the call); err_word: one device word, zeroed
 * once by the owner: sticky, 1 = a bounded hand-off spin gave up (results invalid).  Asynchronous on `stream`. */
typedef struct b2t_wave_t {
  int L, T, B, H;
  const float* gi0;
  const float* w_hh[B2T_MAX_LAYERS_]; const float* b_hh[B2T_MAX_LAYERS_]; const float* w_ih[B2T_MAX_LAYERS_]; const float* b_ih[B2T_MAX_LAYERS_];
  float* h_init[B2T_MAX_LAYERS_]; float* out[B2T_MAX_LAYERS_]; float* outd[B2T_MAX_LAYERS_]; float* reserve[B2T_MAX_LAYERS_];
  const float* dY_top; const float* dh_last; float* dh_init;
  const float* w_hh_t[B2T_MAX_LAYERS_]; const float* w_ih_t[B2T_MAX_LAYERS_]; float* dG[B2T_MAX_LAYERS_];
  float drop_p; uint64_t seed[B2T_MAX_LAYERS_]; long long elem0;
} b2t_wave_t;
int b2t_gru_wave_supported(int L, int T, int B, int H);   /* 0: not held; 1: held (16-unit workgroups); 2: held in the K-split form (H % 128 == 0, H <= 512, a layer's workgroups on one XCD): both passes pay there, the forward one only otherwise */
size_t b2t_gru_wave_ws_bytes(int L, int T, int B, int H, int backward, int dropout);
int b2t_gru_wave_fwd_f32(const b2t_wave_t* d, void* ws, unsigned* err_word, void* stream);
int b2t_gru_wave_bwd_f32(const b2t_wave_t* d, void* ws, unsigned* err_word, void* stream);
int b2t_transpose_f32(const float* in, float* out, int rows, int cols, void* stream);
/* n (1..4) device-to-device copies of 32-bit words in ONE launch: dst[k][0 .. words[k]) = src[k][..] (the static input / output
 * buffers of a replayed streaming graph; no reference counterpart) */
int b2t_copy_segments_b32(const void* const* src, void* const* dst, const long long* words, int n, void* stream);
/* The same with the segment list read when the kernel RUNS, from a device-visible table in pinned host memory: table = {n, src[4],
 * dst[4], words[4]} as 13 int64.  Captured in front of and behind a streaming call's graph, so that a replay serves that call's
 * own input and output tensors (the host rewrites the table; it must not do so while a replay that reads it is in flight). */
int b2t_copy_indirect_b32(const long long* table, int blocks, void* stream);
/* dst[b][0:n] = src[0:n] for b < rows (the learnt initial state h0 broadcast over the batch, rnn_model.py:122-123) */
int b2t_broadcast_rows_f32(const float* src, float* dst, int rows, int n, void* stream);

/* ---- a3-a6 + a8: the model pass as ONE host call (GRUDecoder.forward, rnn_model.py:88-134, and its autograd
 * backward, rnn_trainer.py:547).  The executor owns the HIP streams and events of the execution plan (the L layers are
 * software-pipelined over `chunks` time chunks; the pass is a task graph list-scheduled onto the caller's stream plus
 * three worker streams that the first pipelined pass picks on different command-processor pipes, a one-time ~30 ms
 * measurement that synchronises with the device) and issues every launch of a pass from C++ -- ~250 launches and ~100
 * event operations per training step that cost 14 ms of Python/ctypes time per 25 ms step when they were issued one by
 * one from the host language.  All device memory of a pass still comes from the caller:
 * `ws` (b2t_pass_ws_bytes bytes, any contents; holds the activations forward saves for backward, so one forward/backward
 * pair per ws at a time) and `sync_ws` (b2t_exec_sync_bytes bytes, zeroed once, persistent: the sweeps' counters and
 * error words).  Asynchronous: on return everything is enqueued and the caller's stream has joined the side streams. */
#define B2T_MAX_LAYERS 8
typedef struct b2t_model_t {   /* parameter (or gradient) tensors under the reference's names (rnn_model.py:50-86) */
  int F, H, D, C, L, patch, stride;
  float* day_w; float* day_b;              /* day d at day_w + d*day_w_stride ([F][F]) / day_b + d*day_b_stride ([F]) */
  long long day_w_stride, day_b_stride;
  float* w_ih[B2T_MAX_LAYERS]; float* w_hh[B2T_MAX_LAYERS]; float* b_ih[B2T_MAX_LAYERS]; float* b_hh[B2T_MAX_LAYERS];
  float* out_w; float* out_b; float* h0;
} b2t_model_t;
typedef struct b2t_pass_t {
  int B, T;                  /* batch rows, input frames (T' = (T - patch) / stride + 1 when patch > 0) */
  int chunks;                /* time chunks of the layer pipeline; 1 = layers in sequence on the caller's stream */
  int fwd_mode, bwd_mode;    /* sweep modes (0 / 1, | B2T_GRU_BF16 | B2T_GRU_WIDE) */
  int bf16_gemm;             /* != 0: b2t_gemm_bf16_f32 for every GEMM (the use_amp regime) */
  int save;                  /* != 0: keep what backward needs (gate reserves, layer outputs, U) */
  float in_drop, rnn_drop;   /* dropout probabilities (rnn_model.py:102-103, nn.GRU dropout); 0 in eval */
  uint64_t seed;             /* Philox key of this pass's dropout masks (backward must get the same) */
  int chunks_bwd;            /* time chunks of the backward pass (0: same as `chunks`) */
  int wgrad_chunk_mask;      /* bit l: layer l's weight gradients accumulate chunk by chunk instead of once per layer */
} b2t_pass_t;
typedef struct b2t_exec b2t_exec;
int b2t_exec_create(int n_layers, b2t_exec** out);
int b2t_exec_destroy(b2t_exec* ex);
/* B2T_EXEC_GRAPH=1 (environment, read once): a pass whose every argument repeats (pointers, shapes, modes; the dropout seed when
 * dropout is on; no bucket callback) is built once as a hipGraph from the plan's task graph and replayed with one hipGraphLaunch
 * (EXPERIMENTAL: measured SLOWER than the eager four-queue plan on ROCm 7.0, and 2 of 48 processes ended 3e-6 off the eager loss
 * trajectory -- NOTES.md R4.12).
 * Counters of this executor: graphs built, passes replayed, and whether a build failed (the mode is then off: eager plans). */
int b2t_exec_graph_stats(const b2t_exec* ex, long long* builds, long long* replays, int* failed);
/* HOST ONLY: the list scheduler the executor places a pass's task graph with (HEFT: longest remaining path first, earliest
 * start on one of n_queues in-order queues, gaps may be filled; a dependency that crosses queues costs a fixed hop).  Task i
 * has duration est_us[i], may run on the queues set in qmask[i] (bit q; ignored when n_queues == 1) and depends on
 * deps[dep_off[i] .. dep_off[i+1]) (ids < i: tasks are listed in a topological order).  Outputs: queue[i], the planned
 * start_us[i], and order[0..n) = the order in which the executor issues the tasks (planned start, ties by id).  Pure host
 * arithmetic, deterministic; exported so that the CPU tests can check the schedule's invariants without a GPU. */
int b2t_plan_schedule_host(int n_tasks, const float* est_us, const uint32_t* qmask, const int32_t* dep_off,
                           const int32_t* deps, int n_queues, int32_t* queue, float* start_us, int32_t* order);
/* The same scheduler with the executor's admission control: cls[i] = -1, 0 or 1; tasks of one class (the sweeps whose
 * XCD-local hand-off shares an XCD set) are chained so that the k-th in planned start order waits for the (k-2)-th to
 * finish -- an XCD holds the row groups of two sweeps, and a third one that became partly resident would deadlock all
 * three.  end_us[i] = planned end.  No GPU involved. */
int b2t_plan_admission_host(int n_tasks, const float* est_us, const uint32_t* qmask, const int32_t* dep_off,
                            const int32_t* deps, const int32_t* cls, int n_queues, int32_t* queue, float* start_us,
                            float* end_us, int32_t* order);
size_t b2t_exec_sync_bytes(int n_layers);          /* 2 * n_layers + 1 blocks of b2t_gru_sync_bytes(0): fwd l, then bwd l, then the tile counters of the split-K GEMMs */
size_t b2t_pass_ws_bytes(const b2t_model_t* m, const b2t_pass_t* p);
/* x [B][T][F], day_idx [B], states [L][B][H] or NULL (h0)  ->  logits [B][T'][C], hidden [L][B][H] */
int b2t_model_forward(b2t_exec* ex, const b2t_model_t* prm, const b2t_pass_t* p, const float* x,
                      const int32_t* day_idx, const float* states, float* logits, float* hidden,
                      void* ws, void* sync_ws, void* stream);
/* The STREAMING frame as one launch (BASELINE configs[4]; csrc/stream.hip): GRUDecoder.forward(x, day_idx, states,
 * return_state=True) (model_training/rnn_model.py:88-134) for a handful of patch frames -- day layer, patch, L GRU layers one
 * time step each, head -- in ONE persistent kernel (a workgroup per CU, phases separated by grid barriers, every weight read
 * once), instead of the ~12 dependent launches b2t_model_forward issues for such a call.  Inference only (no dropout, nothing
 * saved), exact fp32, same results as b2t_model_forward up to summation order.
 *   b2t_stream_supported   1 if (model, B, T) can take this path: B <= 64, 1..8 output frames, F and H multiples of 16
 *   b2t_stream_ws_bytes    workspace for one call (any contents)
 *   b2t_stream_forward_f32 x [B][T][F], day_idx [B], states [L][B][H] or NULL (h0) -> logits [B][T'][C], hidden [L][B][H];
 *                          sync: b2t_stream_sync_bytes() bytes, zeroed once, persistent (barrier counters, re-armed by every
 *                          call; word 2 = sticky error: a barrier that timed out -- the launch could not become fully
 *                          resident -- poisons logits[0] with NaN; the last 64 words = phase stamps of workgroup 0 on the
 *                          100 MHz clock, for profiling).  Asynchronous on `stream`. */
int b2t_stream_supported(const b2t_model_t* prm, int B, int T);
size_t b2t_stream_ws_bytes(const b2t_model_t* prm, int B, int T);
size_t b2t_stream_sync_bytes(void);
int b2t_stream_forward_f32(const b2t_model_t* prm, int B, int T, const float* x, const int32_t* day_idx, const float* states,
                           float* logits, float* hidden, void* ws, size_t ws_bytes, void* sync, void* stream);
/* Gradients of every parameter given dlogits [B][T'][ldd] (ldd % 4 == 0, >= C), written (not accumulated) into `grd`;
 * days absent from day_idx are not touched.  dhidden [L][B][H] optional gradient wrt the final states; dstates
 * [L][B][H] optional output = gradient wrt the initial states (when the forward got `states`, h0 receives no gradient).
 * bucket_cb(user, id, stream) is called right after the last launch of a gradient bucket was enqueued on `stream`
 * (ids: 0 = head, 1 + l = GRU layer l, L + 1 = h0, L + 2 = day layers) -- the data-parallel reducer hooks its
 * all-reduce there, in backward-completion order.  `p`, x, day_idx, ws: as given to the forward. */
typedef void (*b2t_bucket_cb)(void* user, int bucket, void* stream);
int b2t_model_backward(b2t_exec* ex, const b2t_model_t* prm, const b2t_model_t* grd, const b2t_pass_t* p,
                       const float* x, const int32_t* day_idx, const float* dlogits, int ldd,
                       const float* dhidden, float* dstates, int custom_states, void* ws, void* sync_ws,
                       b2t_bucket_cb bucket_cb, void* user, void* stream);
/* Live per-launch timing (HIP events on the stream each kernel is launched on) for bench.py's roofline: while on, every
 * GEMM / sweep launch of the two calls above is bracketed; b2t_exec_profile_read synchronises the device and returns up
 * to `cap` records {kind, flops, milliseconds} (kind: 0-3 gemm_f32<a_kcontig,b_kcontig>, 4-7 gemm_bf16<..>, 8 forward
 * sweep, 9 backward sweep) and clears them.  Returns the number of records, < 0 on error. */
int b2t_exec_profile(b2t_exec* ex, int on);
int b2t_exec_profile_read(b2t_exec* ex, int* kind_host, double* flops_host, float* ms_host, int cap);

/* ---- a7: log-softmax + CTC loss (torch.nn.CTCLoss(blank=0,'none') at rnn_trainer.py:242,538-545)
 * logits [B][T][C] batch-first.  targets [B][S_max] int32 (0-padded), in_len/tgt_len [B] int32.
 * loss [B] = -log p(target | logits[:in_len]) (inf when infeasible; zero_infinity=False).
 * alpha_ws: scratch of B*T*(2*S_max+1) floats for the loss alone, TWICE that when dlogits is requested
 * (alpha rows, then beta rows: the two recursions run concurrently and a third pass forms the gradient).
 * dlogits [B][T][ldd] (ldd >= C) = grad_scale * d loss_b / d logits (grad_scale = 1/B for the
 * reference's torch.mean, rnn_trainer.py:545); exactly 0 for t >= in_len[b]. dlogits may be NULL. */
int b2t_ctc_loss_f32(const float* logits, const int32_t* targets, const int32_t* in_len,
                     const int32_t* tgt_len, float* loss, float* alpha_ws, float* dlogits,
                     int B, int T, int C, int S_max, int ldd, float grad_scale, void* stream);

/* ---- a9/a10: gradient clipping + AdamW on a flat parameter arena (no host sync) ----------------
 * The host keeps every parameter in ONE fp32 arena, each tensor ("segment") padded to a multiple of
 * 1024 floats; chunk2seg[c] (int32[nchunks]) names the tensor of 1024-float chunk c.  Device tables
 * per segment: seg_group (0 bias / 1 day / 2 other, rnn_trainer.py:267-269), seg_day (day index or -1),
 * seg_step (the tensor's own AdamW step count), active (has a gradient this step).
 * b2t_opt_prepare: active[s] = seg_day[s] < 0 || seg_day[s] in day_idx[0..B)   — days absent from the
 *   batch have grad None in the reference (rnn_trainer.py:514) and are skipped by clip and AdamW.
 * b2t_grad_norm_clip_f32 (clip_grad_norm_, rnn_trainer.py:551-555): out4 = {sum g^2, norm,
 *   clip_coef = min(1, max_norm/(norm+1e-6)), status} over active tensors (max_norm <= 0: coef 1); also advances
 *   seg_step of active tensors (seg_step may be NULL).  partial_ws: nchunks floats.
 *   status (sticky: once non-zero it stays): 0 = OK; 1 = a persistent sweep reported a hand-off timeout (any of the
 *   n_err error words err_words[i * err_stride] is set: the gradients are invalid); 2 = non-finite gradient norm (the
 *   reference raises there: error_if_nonfinite=True, rnn_trainer.py:553).  With status != 0 the step counters do not
 *   advance and b2t_adamw_f32 (given out4 as `clip4`) leaves parameters and moments untouched: a bad step is never
 *   applied; the host raises when it next reads out4.  err_words may be NULL. */
int b2t_opt_prepare(const int32_t* day_idx, int B, const int32_t* seg_day, int nseg, int32_t* active,
                    void* stream);
int b2t_grad_norm_clip_f32(const float* grads, const int32_t* chunk2seg, const int32_t* active,
                           int nchunks, float max_norm, float* partial_ws, float* out4,
                           int32_t* seg_step, int nseg, const uint32_t* err_words, int n_err,
                           long long err_stride, void* stream);
/* Data parallel only (no reference counterpart: the reference is single-process): the step counters
 * advance AFTER the ranks have MAX-reduced out4[3], so that a step one rank refuses is refused -- counters included -- by
 * all: call b2t_grad_norm_clip_f32 with seg_step = NULL, all-reduce out4[3], then this (advances the counters of active
 * tensors iff out4[3] == 0), then b2t_adamw_f32. */
int b2t_opt_advance(const int32_t* active, int32_t* seg_step, int nseg, const float* out4, void* stream);
/* AdamW (torch.optim.AdamW, rnn_trainer.py:283-290), active tensors only, k = seg_step (1-based):
 *   g *= clip4[2] (if apply_clip); p *= 1-lr*wd; m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
 *   p -= lr/(1-b1^k) * m / (sqrt(v)/sqrt(1-b2^k) + eps)      (bias corrections evaluated in fp64)
 * lr3_host / wd3_host: HOST arrays of 3 floats (per group: bias, day, other).  clip4 = the out4 of
 * b2t_grad_norm_clip_f32 (may be NULL): nothing is updated when clip4[3] != 0. */
int b2t_adamw_f32(float* params, float* grads, float* exp_avg, float* exp_avg_sq,
                  const int32_t* chunk2seg, const int32_t* active, const int32_t* seg_group,
                  const int32_t* seg_step, int nchunks, const float* clip4, int apply_clip,
                  const float* lr3_host, const float* wd3_host, double beta1, double beta2, float eps,
                  void* stream);

/* ---- a12: greedy CTC decode (rnn_trainer.py:724-728) --------------------------------------
 * argmax over classes (first max wins, like torch.argmax), collapse repeats, drop blank.
 * logits [B][T][C]; lens [B]; out_ids [B][T] int32; out_len [B] int32; argmax_out [B][T] optional. */
int b2t_greedy_decode_f32(const float* logits, const int32_t* lens, int32_t* out_ids,
                          int32_t* out_len, int32_t* argmax_out, int B, int T, int C, void* stream);
/* Levenshtein distance per sentence (torchaudio.functional.edit_distance, rnn_trainer.py:734).
 * a [B][La_max], b [B][Lb_max] int32, lengths, dist [B] int32. */
int b2t_edit_distance_i32(const int32_t* a, const int32_t* a_len, int La_max, const int32_t* b,
                          const int32_t* b_len, int Lb_max, int32_t* dist, int B, void* stream);

/* ---- a14/a17: LM-decoder prologue + CTC prefix beam search ---------------------------------
 * lm_decoder.cc:14-37 DecodeNumpy: logp = log_softmax(logits) - log_priors; logp[:,0] -= blank_penalty.
 * logits/logp [rows][C]; log_priors may be NULL (zeros). */
int b2t_lm_prologue_f32(const float* logits, const float* log_priors, float blank_penalty,
                        float* logp, int rows, int C, void* stream);
/* Batched CTC prefix beam search — the LM-free searcher BrainSpeechDecoder uses when no TLG graph is loaded
 * (language_model/runtime/core/decoder/ctc_prefix_beam_search.cc:44-136; PrefixScore ctc_prefix_beam_search.h:27-42;
 * LogAdd language_model/runtime/core/utils/utils.cc:24-30).  One workgroup per utterance, beam in LDS,
 * prefix trie + search state in the caller-provided `state` block (U * b2t_beam_state_bytes bytes), which
 * persists between calls so logp can be fed chunk by chunk (streaming).  b2t_beam_reset = Reset().
 *   logp [U][T][C] (C <= 64), lens [U] (NULL: T), first_beam <= 16 classes per frame, second_beam <= 128 prefixes.
 * Outputs, sorted best first (slots beyond the live beam: hyp_len = -1):
 *   hyps [U][second_beam][max_len] token ids, hyp_len [U][second_beam], score = LogAdd(s, ns), vscore = Viterbi
 *   score, times [U][second_beam][max_len] (frame of each token on the Viterbi path; may be NULL).  Only the first
 *   hyp_len entries of a hyps / times row are written; the rest keeps whatever the buffer held.
 * max_nodes bounds the trie (<= second_beam new nodes per frame); b2t_beam_overflowed reports exhaustion. */
size_t b2t_beam_state_bytes(int max_len, int max_nodes);
int b2t_beam_reset(void* state, int U, int max_len, int max_nodes, void* stream);
int b2t_prefix_beam_search_f32(const float* logp, const int32_t* lens, int U, int T, int C, int first_beam,
                               int second_beam, int blank, void* state, int max_len, int max_nodes,
                               int32_t* hyps, int32_t* hyp_len, float* score, float* vscore, int32_t* times,
                               void* stream);
int b2t_beam_overflowed(const void* state, int U, int max_len, int max_nodes, int* flag_host, void* stream);
/* The same search with a token-level back-off n-gram LM fused in (the "CTC prefix beam + n-gram" decode of
 * BASELINE.json configs 4-5; the reference reaches its ARPA LMs through a word-level WFST,
 * language_model/runtime/core/decoder/ctc_wfst_beam_search.cc, whose graphs are not in the checkout: pinned by
 * oracle/b2t_oracle.py:prefix_beam_search_lm only).  The LM is an automaton in device memory built by
 * nejm-brain-to-text_amd/ngram_lm.py from ARPA text: lm_child [n_nodes][lm_vocab] (-1 = absent), lm_logp / lm_bow
 * [n_nodes] natural logs, lm_suffix / lm_nstate [n_nodes]; vocabulary = class ids 0..C-1, then <s>, </s>, <unk>.
 * Every emitted token adds alpha * ln p(token | history) + beta; pruning and ranking use CTC score + LM score.
 * lm_score [U][second_beam] receives the LM part (plus alpha * ln p(</s> | history) when lm_eos >= 0); `score`
 * stays the CTC part.  Same state block, streaming and reset rules as b2t_prefix_beam_search_f32. */
int b2t_prefix_beam_search_lm_f32(const float* logp, const int32_t* lens, int U, int T, int C, int first_beam,
                                  int second_beam, int blank, void* state, int max_len, int max_nodes,
                                  int32_t* hyps, int32_t* hyp_len, float* score, float* vscore, int32_t* times,
                                  const int32_t* lm_child, const float* lm_logp, const float* lm_bow,
                                  const int32_t* lm_suffix, const int32_t* lm_nstate, int lm_vocab,
                                  int lm_start_state, int lm_eos, float alpha, float beta, float unk_logp,
                                  float* lm_score, void* stream);

/* Word-level variant: the search is constrained by a pronunciation lexicon (trie over the classes; words are
 * delimited by the SIL class, evaluate_model_helpers.py:79-83 class 1) and a WORD n-gram is fused in: closing a word
 * with SIL (or the end of the utterance) adds alpha * ln p(word | word history) + beta, homophones resolve to the
 * word the LM prefers.  Stands in for the TLG graph search of the reference (ctc_wfst_beam_search.cc:70-160 over
 * T o L o G built by tools/fst/make_tlg.sh:29-46) -- without its graph optimisations and optional-silence arcs, and
 * pinned by the oracle only (oracle/b2t_oracle.py:prefix_beam_search_lexicon).  Tables: ngram_lm.Lexicon /
 * ngram_lm.SparseNGramLM (sorted child arrays, bisection).  lm_score is -inf for hypotheses ending inside a word. */
typedef struct {
  const int32_t* lex_child;   /* [n_lex_nodes][C], -1 = no edge */
  const int32_t* lex_wbeg;    /* [n_lex_nodes] */
  const int32_t* lex_wend;    /* [n_lex_nodes]: words ending at the node = wlist[wbeg..wend) */
  const int32_t* wlist;
  const int32_t* lm_cb;       /* [n_lm_nodes] children of LM node n: (lm_ctok, lm_cnode)[cb..ce), sorted by word id */
  const int32_t* lm_ce;
  const int32_t* lm_ctok;
  const int32_t* lm_cnode;
  const float* lm_logp;       /* [n_lm_nodes] natural logs */
  const float* lm_bow;
  const int32_t* lm_suffix;
  const int32_t* lm_nstate;
  int32_t lm_start_state, lm_eos /* word id of </s>, or -1 */, sil;
  float alpha, beta, unk_logp;
} b2t_lexlm_t;
int b2t_prefix_beam_search_lex_f32(const float* logp, const int32_t* lens, int U, int T, int C, int first_beam,
                                   int second_beam, int blank, void* state, int max_len, int max_nodes,
                                   int32_t* hyps, int32_t* hyp_len, float* score, float* vscore, int32_t* times,
                                   const b2t_lexlm_t* d, float* lm_score, void* stream);

/* ---- f4: decode-graph compiler on the host (csrc/graphc.cpp; no device work, no stream) ------------------------------------
 * The FST algebra of the reference's graph recipe, language_model/tools/fst/make_tlg.sh:29-46
 *   fsttablecompose L.fst G.fst | fstdeterminizestar --use-log=true | fstminimizeencoded | fstarcsort --sort_type=ilabel > LG.fst
 *   fsttablecompose T.fst LG.fst > TLG.fst
 * over handles to host FSTs in CSR form (tropical weights = costs; +inf final cost = not final).  A handle is owned by the
 * caller (b2t_fst_free); every operation returns a NEW handle or NULL (b2t_last_error()).
 *   b2t_fst_from_arrays     arcs in any order (src, ilabel, olabel, weight, dst); arc order within a state is kept
 *   b2t_fst_info            {n_states, n_arcs, start, n_final}
 *   b2t_fst_to_arrays       row[n_states + 1] (int64), ilabel / olabel / weight / next [n_arcs], final_cost [n_states]
 *   b2t_fst_compose         a o b (a's olabels against b's ilabels), epsilon-matching filter: fst::Compose / fsttablecompose
 *   b2t_fst_trim            fstconnect: accessible and co-accessible states only; start state 0
 *   b2t_fst_determinize_star  kaldi/fstext/determinize-star-inl.h (epsilon removal + determinisation on the input side, output
 *                           strings as chains of input-epsilon arcs); use_log: weights of merged paths are log-added
 *                           (--use-log=true), else the minimum; delta: weight tolerance of subset equality (Kaldi 1/1024);
 *                           max_states <= 0: unlimited.  Fails on non-functional input (needs the lexicon's disambiguation
 *                           symbols, tools/fst/add_lex_disambig.pl)
 *   b2t_fst_minimize_encoded  kaldi/fstext/fstext-utils.h:110-116: weights quantised to multiples of delta, (ilabel, olabel,
 *                           weight) as one label, acceptor minimisation
 *   b2t_fst_arcsort         fstarcsort --sort_type=ilabel (by_olabel = 0) / olabel (1), stable
 *   b2t_fst_read_openfst / b2t_fst_write_openfst   OpenFST binary container, fst type "vector", arc type "standard"
 *   b2t_fst_grammar_score   cost of a word-id sequence through an ilabel-sorted grammar whose back-off arcs carry `backoff_label`
 *                           (what composing a lattice path with G adds: brain_speech_decoder.cc:44-58); +inf = not accepted */
void* b2t_fst_from_arrays(int n_states, int start, long long n_arcs, const int32_t* src, const int32_t* ilabel, const int32_t* olabel,
                          const float* weight, const int32_t* dst, const float* final_cost);
void b2t_fst_free(void* fst);
int b2t_fst_info(const void* fst, long long* out4);
int b2t_fst_to_arrays(const void* fst, long long* row, int32_t* ilabel, int32_t* olabel, float* weight, int32_t* next, float* final_cost);
void* b2t_fst_compose(const void* a, const void* b);
void* b2t_fst_trim(const void* fst);
void* b2t_fst_determinize_star(const void* fst, int use_log, float delta, long long max_states);
void* b2t_fst_minimize_encoded(const void* fst, float delta);
void* b2t_fst_arcsort(const void* fst, int by_olabel);
void* b2t_fst_read_openfst(const char* path);
int b2t_fst_write_openfst(const void* fst, const char* path);
double b2t_fst_grammar_score(const void* fst, const int32_t* words, int n_words, int backoff_label);
/* fst::ReadAndPrepareLmFst (kaldi/fstext/kaldi-fst-io.cc:129-147): NEW handle = the grammar projected on its output labels if it is
 * not an acceptor, arc-sorted by ilabel; *backoff_label = the label its back-off arcs carry (0; `disambig_id` for an acceptor with #0
 * on both sides and no label-0 arc; disambig_id < 0: never). */
void* b2t_fst_prepare_lm(const void* fst, int disambig_id, int* backoff_label);

/* ---- a15/a16: WFST token passing (the reference's LM decode proper) ---------------------------------------------------
 * CtcWfstBeamSearch::Search / FinalizeSearch (language_model/runtime/core/decoder/ctc_wfst_beam_search.cc:70-160) over
 * kaldi's LatticeFasterDecoder (language_model/runtime/core/kaldi/decoder/lattice-faster-decoder.cc: ProcessEmitting
 * :722-824, ProcessNonemitting :839-909, GetCutoff :650-720, FindOrAddToken :250-295, FinalizeDecoding :632-647), batched:
 * one workgroup per utterance, the decode graph (T o L o G, nejm-brain-to-text_amd/wfst.py) in device memory as CSR arrays
 * shared by all utterances, each frame's token hash in LDS, tokens and forward links appended to the utterance's state
 * block (U * b2t_wfst_state_bytes bytes; persists between calls: logp may be fed chunk by chunk).
 * Graph arrays: row [n_states + 1], arcs sorted by ilabel within a state (input-epsilon arcs first, n_eps[s] of them);
 * ilabel 0 = epsilon, ilabel i > 0 reads acoustic_scale * logp[i - 1] (DecodableTensorScaled, :27-33); final_cost +inf =
 * not final.  Options: LatticeFasterDecoderConfig + CtcWfstBeamSearchOptions (production values:
 * language-model-standalone.py:486-496 -- beam 17, max_active 7000, min_active 200, lattice_beam 8, acoustic_scale
 * 0.325, blank_skip_thresh 1.0); max_frames / max_tokens / max_links / hash_size (power of two; <= 16384 keeps the hash in
 * LDS) are per-utterance capacities -- exhaustion sets the overflow word (header word 3) and invalidates the result. */
typedef struct {
  const int32_t* row; const int32_t* ilabel; const int32_t* olabel; const float* weight; const int32_t* next;
  const int32_t* n_eps; const float* final_cost; int32_t n_states, start;
  /* compact arcs (compact != 0): 10 bytes per arc instead of 16 -- labels[a] = ilabel | olabel << 7 (ilabel <= 127, olabel <
   * 2^25), weight_f16[a] = the weight as IEEE binary16 (relative error <= 2^-11), next as above; ilabel / olabel / weight are
   * then not read and may be NULL.  A search on a compact graph equals the search on the full-width graph whose weights were
   * rounded to binary16 beforehand, bit for bit.  No reference counterpart (OpenFST's ConstFst keeps 16-byte arcs). */
  const uint32_t* labels; const uint16_t* weight_f16; int32_t compact;
} b2t_wfst_graph_t;
typedef struct {
  float beam, lattice_beam, beam_delta, acoustic_scale, length_penalty, blank_skip_thresh;
  int32_t max_active, min_active;
  int32_t max_frames, max_tokens, max_links, hash_size;
} b2t_wfst_opts_t;
size_t b2t_wfst_state_bytes(int max_frames, int max_tokens, int max_links, int hash_size);
/* Workgroups per utterance b2t_wfst_search_f32 will use for U utterances: 8 / 4 / 2 share one utterance's frame (a cluster
 * behind one XCD's L2, cluster barriers and L2 atomics) while every cluster fits the chip, else 1 (the same kernel with one
 * member per utterance).  No reference counterpart (the reference's search is one CPU thread). */
int b2t_wfst_cluster_size(int U);
int b2t_wfst_set_cluster(int G);   /* process-wide override: 2 / 4 / 8 / 16 / 32 workgroups per utterance; 1 = the single-workgroup kernel (frame hash in LDS; the form the cluster search is tested against); -1 = the cluster kernel with one member; 0 = automatic (B2T_WFST_CLUSTER) */
int b2t_wfst_reset(const b2t_wfst_graph_t* g, const b2t_wfst_opts_t* o, void* state, int U, void* stream);   /* InitDecoding */
/* logp [U][T][C] (C <= 64), lens [U] or NULL: blank-frame skipping + AdvanceDecoding(.., 1) per kept frame */
int b2t_wfst_search_f32(const b2t_wfst_graph_t* g, const b2t_wfst_opts_t* o, void* state, const float* logp,
                        const int32_t* lens, int U, int T, int C, void* stream);
/* Best path by backpointers (lattice-faster-online-decoder.cc:58-150): alignment [U][max_len] (graph ilabels) with the
 * input frame of each entry, words [U][max_len] (olabels), costs [U][2] = {graph (+ final), acoustic}.
 * use_final = 0: partial result (Search's GetBestPath(.., false)); 1: after b2t_wfst_finalize. */
int b2t_wfst_best_path(const b2t_wfst_graph_t* g, const b2t_wfst_opts_t* o, const void* state, int U, int use_final,
                       int max_len, int32_t* alignment, int32_t* align_frame, int32_t* n_align, int32_t* words,
                       int32_t* n_words, float* costs, void* stream);
/* FinalizeDecoding: final costs + backward pruning with lattice_beam; marks the surviving forward links. */
int b2t_wfst_finalize(const b2t_wfst_graph_t* g, const b2t_wfst_opts_t* o, void* state, int U, void* stream);
/* PruneActiveTokens(delta) (lattice-faster-decoder.cc:516-545; the reference calls it every prune_interval = 25 decoded
 * frames with delta = lattice_beam * prune_scale, :592-630): call between two b2t_wfst_search_f32 calls.  Prunes forward links
 * and tokens of every frame but the newest against the best path so far and COMPACTS the utterance's token / link arrays, so
 * a streamed utterance holds its pruned lattice plus the frames since the last call.  Never changes the final lattice.
 * min_fill in [0, 1]: an utterance whose token AND link arrays are filled below that fraction of their capacity skips the
 * pass (the pass only bounds memory); 0 = always prune, the reference's behaviour. */
int b2t_wfst_prune(const b2t_wfst_graph_t* g, const b2t_wfst_opts_t* o, void* state, int U, float delta, float min_fill,
                   void* stream);
/* The pruned lattice in compact form, after b2t_wfst_finalize (GetRawLattice, lattice-faster-decoder.cc:106-186): surviving
 * tokens renumbered, surviving links as arcs with acoustic = link acoustic cost - the frame's cost offset, final costs of
 * the last frame's tokens.  Per utterance u: arcs at [u * cap_arcs ..), finals at [u * cap_final ..),
 * counts[5u ..] = {n_states, n_arcs, n_final, start state, 1 if a capacity was too small}. */
int b2t_wfst_lattice(const b2t_wfst_graph_t* g, const b2t_wfst_opts_t* o, void* state, int U, int cap_arcs, int cap_final,
                     int32_t* counts, int32_t* src, int32_t* dst, int32_t* ilabel, int32_t* olabel, float* graph,
                     float* acoustic, int32_t* final_state, float* final_cost, void* stream);
/* Byte offsets of the arrays inside ONE utterance's state block, for copying the pruned lattice out:
 * off16 = {header, mapping, tok_off, link_off, cost_offset, tok_state, tok_cost (order-preserving u32 of the f32 cost),
 * tok_extra, link_src, link_dst, link_arc, link_ac, link_graph, link_alive (u8), tok_best, last_prob}. */
int b2t_wfst_state_offsets(int max_frames, int max_tokens, int max_links, int hash_size, long long* off16);
/* HOST function: the n-best distinct word sequences of a pruned lattice (what GetLattice's DeterminizeLatticePruned +
 * ShortestPath(nbest) yield, ctc_wfst_beam_search.cc:138-143): arcs (src, dst, ilabel, olabel, graph, acoustic) over
 * n_states lattice states, finals (state, cost).  Outputs: for entry k the words out_words[w_off[k] .. w_off[k+1]), the
 * alignment out_ali[a_off[k] .. a_off[k+1]) (ilabels of its best path) and costs[2k] = graph (+ final), costs[2k+1] =
 * acoustic.  Returns the number of entries (<= nbest), < 0 on error (buffers too small: -2). */
int b2t_lattice_nbest_host(int n_states, int start, int n_arcs, const int32_t* src, const int32_t* dst,
                           const int32_t* ilabel, const int32_t* olabel, const float* graph, const float* acoustic,
                           int n_final, const int32_t* final_state, const float* final_cost, int nbest, float beam,
                           int32_t* out_words, int32_t* w_off, int w_cap, int32_t* out_ali, int32_t* a_off, int a_cap,
                           float* costs);
/* HOST function: BrainSpeechDecoder::Rescore (brain_speech_decoder.cc:47-101 -- LatticeRescore with the graph's grammar at
 * scale -1, then with the rescoring grammar at scale +1, then ShortestPath(n)) on the same pruned lattice: every word sequence
 * W of the lattice gets graph(W) - G_old(W) + G_new(W) (each the cheapest route through the grammar, back-off arcs -- those
 * carrying `backoff_label` on the input side -- free to take anywhere; final costs included), by composing the lattice with
 * both grammars determinised on the fly; the n-best distinct word sequences are ranked by the NEW total cost among the
 * sequences whose OLD cost is within `beam` of the best OLD cost (the contents of the reference's lat_).  g_old / g_new:
 * b2t_fst_* handles, arc-sorted by ilabel.  Outputs as b2t_lattice_nbest_host (costs[2k] = the exchanged graph cost);
 * stats4 (optional) = {product states, product arcs, determinised states of g_old, of g_new}.
 * Threads: re-entrant (callers run one call per utterance on a pool); work arrays are kept per calling thread.  A lattice of
 * >= 60000 arcs is determinised with helper threads of its own for the duration of the call (4 including the caller;
 * B2T_RESCORE_BIG_THREADS=n, 1 = none; B2T_RESCORE_THREADS=n forces n for every lattice) -- the result does not depend on it. */
int b2t_lattice_rescore_nbest_host(int n_states, int start, int n_arcs, const int32_t* src, const int32_t* dst,
                                   const int32_t* ilabel, const int32_t* olabel, const float* graph, const float* acoustic,
                                   int n_final, const int32_t* final_state, const float* final_cost,
                                   const void* g_old, const void* g_new, int backoff_label, int nbest, float beam,
                                   int32_t* out_words, int32_t* w_off, int w_cap, int32_t* out_ali, int32_t* a_off, int a_cap,
                                   float* costs, long long* stats4);
/* HOST function: CtcWfstBeamSearch::ConvertToInputs (ctc_wfst_beam_search.cc:162-188) for the n alignments
 * ali[a_off[k] .. a_off[k+1]) that b2t_lattice_nbest_host returned: blanks (ilabel 1) dropped, repeats merged, ilabel - 1;
 * the time of a unit is the frame of its LAST repeated label -- mapping[position] (decoded frame -> input frame, F entries)
 * when the alignment covers all F decoded frames, else the position itself.  Entry k: out_inputs / out_times
 * [out_off[k] .. out_off[k+1]).  Returns 0, or -2 when `cap` is too small. */
int b2t_nbest_convert_to_inputs(const int32_t* ali, const int32_t* a_off, int n, const int32_t* mapping, int F,
                                int32_t* out_inputs, int32_t* out_times, int32_t* out_off, int cap);

#ifdef __cplusplus
}
#endif
#endif /* B2T_H */
