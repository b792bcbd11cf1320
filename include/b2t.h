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

#define B2T_VERSION 1

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
 * epilogue: 0 = none, 1 = softsign(v) (rnn_model.py:47,99).  accumulate: C += result. */
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
   * C + z*c_sz + ks*c_ks; the caller sums the slabs deterministically (b2t_colsum_f32). */
  int splitk; long long c_ks;
  /* gap in A's contiguous index i (k if a_kcontig, else m): for i >= a_brk the element is read at i + a_gap.
   * Lets dGi = dG[:, 0:2H] ++ dG[:, 3H:4H] be ONE operand (a_brk = 2H, a_gap = H).  a_brk = 0: off; must be a
   * multiple of 16 (k) / 128 (m, with M % 128 == 0). */
  int a_brk; int a_gap;
} b2t_gemm_desc;
int b2t_gemm_f32(const b2t_gemm_desc* d, void* stream);
/* The same GEMM with the operands rounded to bf16 (nearest-even) on their way to the matrix cores, fp32 accumulation and
 * fp32 output: the matmul regime of the reference's `use_amp` / autocast(bfloat16) training (rnn_trainer.py:535).  Tensors
 * stay fp32 in memory.  Split-K chunks and a_brk along k are multiples of 32 here. */
int b2t_gemm_bf16_f32(const b2t_gemm_desc* d, void* stream);

/* ---- elementwise helpers ------------------------------------------------------------------
 * softsign backward (rnn_model.py:99 autograd): du[i] *= (1-|u[i]|)^2, in place. */
int b2t_softsign_bwd_f32(const float* u, float* du, long long n, void* stream);
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
/* dropout (rnn_model.py:102-103, nn.GRU inter-layer dropout :70): y = x*mask/(1-p), Philox mask
 * keyed by (seed, elem0 + index) so that a tensor processed in chunks gets the mask of the whole tensor;
 * forward and backward apply the same mask. In place allowed. */
int b2t_dropout_f32(const float* x, float* y, long long n, float p, uint64_t seed, long long elem0,
                    void* stream);
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
 * W_hh slices resident in registers, agent-scope flag hand-off of h_t between workgroups),
 * 2 = persistent forward with data-tagged granules (backward: as mode 1), 3 = software-pipelined
 * persistent sweep, 4 row groups per workgroup (csrc/gru_pipeline.hip; shapes it does not cover --
 * fewer than 3 row groups, H > 512 backward / > 768 forward -- run as mode 1).  All modes produce the
 * same out / reserve / dG layouts and can be mixed between calls on one workspace.
 * mode 1 | B2T_GRU_BF16: the recurrent product takes bf16 operands (h_{t-1} / dG_{t+1} and the W_hh slice rounded to
 * nearest-even bf16, v_mfma_f32_16x16x16_bf16, fp32 accumulate; gates and everything stored stay fp32): the reference's
 * autocast(bfloat16) regime for the GRU, opt-in.
 * sync_ws: device scratch of b2t_gru_sync_bytes(T) bytes (persistent mode; zeroed by the call). */
#define B2T_GRU_BF16 0x100
#define B2T_GRU_WIDE 0x200   /* with B2T_GRU_BF16: 32 hidden units per workgroup (half the workgroups per sweep), H % 32 == 0, H <= 512 */
size_t b2t_gru_sync_bytes(int T);
/* Workspace size valid for every mode (mode 2 = persistent sweep with data-tagged 8-byte {value,tag}
 * granule hand-off, csrc/gru_granule.hip: needs T*B*H*8 bytes of granules behind the control words). */
size_t b2t_gru_ws_bytes(int T, int B, int H);
/* Persistent mode only: copies the sweep's error word to the host and synchronises the stream.
 * *status_host = 0: clean; 1: a bounded hand-off spin gave up (results of that sweep are invalid). */
int b2t_gru_sync_status(const void* sync_ws, int T, int B, int* status_host, void* stream);
int b2t_gru_layer_fwd_f32(const float* gi, const float* w_hh, const float* b_hh,
                          const float* h_init, float* out, float* reserve, float* h_last,
                          int T, int B, int H, int mode, void* sync_ws, void* stream);
/* Backward sweep.  dY [T][B][H] grad wrt this layer's outputs (plus dh_last
 * [B][H] optional grad wrt the final state).  w_hh_t [H][3H] is W_hh transposed (b2t_transpose).
 * dG [T][B][4H] = (dr_pre, dz_pre, dn_pre*r, dn_pre): dGh = cols [0,3H), dGi = cols [0,2H)+[3H,4H).
 * dh_init [B][H] = carry after t=0.  carry_ws: [B][H] floats scratch. */
int b2t_gru_layer_bwd_f32(const float* dY, const float* dh_last, const float* reserve,
                          const float* out, const float* h_init, const float* w_hh_t,
                          float* dG, float* dh_init, float* carry_ws,
                          int T, int B, int H, int mode, void* sync_ws, void* stream);
int b2t_transpose_f32(const float* in, float* out, int rows, int cols, void* stream);

/* Persistent sweeps (mode 1) released sub-chunk by sub-chunk.  A launch of T steps is cut into sub-chunks of `sub`
 * steps (forward: in time order; backward: counted from the END of the launch, the order the sweep visits them).
 * The kernel enters sub-chunk k only once ready[k] >= epoch (the caller writes it with b2t_stream_write_value32 behind
 * the GEMM that produced gi / dY of that sub-chunk) and stores done[k] = epoch once every workgroup has finished it
 * (out / dG of the sub-chunk are then in memory; wait for it with b2t_stream_wait_value32_gte in front of the consuming
 * GEMM).  ready / done: plain device words (NULL = no waiting / no signalling); epoch must grow from pass to pass. */
int b2t_gru_layer_fwd_flagged_f32(const float* gi, const float* w_hh, const float* b_hh, const float* h_init,
                                  float* out, float* reserve, float* h_last, int T, int B, int H, void* sync_ws,
                                  const uint32_t* ready, uint32_t* done, int sub, uint32_t epoch, void* stream);
int b2t_gru_layer_bwd_flagged_f32(const float* dY, const float* dh_last, const float* reserve, const float* out,
                                  const float* h_init, const float* w_hh_t, float* dG, float* dh_init,
                                  int T, int B, int H, void* sync_ws,
                                  const uint32_t* ready, uint32_t* done, int sub, uint32_t epoch, void* stream);
/* ---- a5 (mode 4): the whole GRU stack as ONE persistent launch (nn.GRU forward, rnn_model.py:126, all layers).
 * Layer 0 reads its input projection gi0 [T][B][3H] (b_ih folded in) from a GEMM that ran before; layers l >= 1 form
 * x_t W_ih^T inside the sweep from the tile layer l-1 has just published.  out[l] is [T+1][B][H]: slab 0 holds the
 * initial state on entry, slab t+1 receives h_t.  nn.GRU inter-layer dropout: when drop_mask[l] ([T][B][H] factors, 0 or
 * 1/(1-p), e.g. from b2t_dropout_mask_f32) and out_drop[l] are given, layer l also writes h_t * mask to out_drop[l] (same
 * layout as out[l]) and layer l+1 consumes that copy.  reserve[l]: [T][B][4H] or NULL.
 * sync_ws: b2t_gru_sync_bytes() bytes, zeroed once, one per concurrent call.  Returns 4 (and launches nothing) when the
 * shape is not covered -- (H/16) * L * ceil(B/64) workgroups must be resident at once, H <= 512 -- the caller then runs
 * the per-layer sweeps. */
#define B2T_STACK_MAX_LAYERS 8
typedef struct {
  int T, B, H, L;
  const float* gi0;
  const float* w_hh[B2T_STACK_MAX_LAYERS];
  const float* w_ih[B2T_STACK_MAX_LAYERS];   /* [0] unused */
  const float* b_hh[B2T_STACK_MAX_LAYERS];
  const float* b_ih[B2T_STACK_MAX_LAYERS];   /* [0] unused */
  float* out[B2T_STACK_MAX_LAYERS];
  float* out_drop[B2T_STACK_MAX_LAYERS];     /* NULL: no dropped copy for that layer */
  const float* drop_mask[B2T_STACK_MAX_LAYERS];   /* NULL together with out_drop */
  float* reserve[B2T_STACK_MAX_LAYERS];
  int bf16;   /* != 0: bf16 operands for both products (as B2T_GRU_BF16 for the per-layer sweeps) */
} b2t_gru_stack_t;
int b2t_gru_stack_fwd_f32(const b2t_gru_stack_t* d, void* sync_ws, void* stream);

/* Stream-ordered 32-bit word write / wait-until->= executed by the command processor (no kernel launch). */
int b2t_stream_write_value32(void* ptr, uint32_t value, void* stream);
int b2t_stream_wait_value32_gte(void* ptr, uint32_t value, void* stream);

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
 * b2t_grad_norm_clip_f32 (clip_grad_norm_, rnn_trainer.py:551-555): out3 = {sum g^2, norm,
 *   clip_coef = min(1, max_norm/(norm+1e-6))} over active tensors (max_norm <= 0: coef 1); also advances
 *   seg_step of active tensors (seg_step may be NULL).  partial_ws: nchunks floats. */
int b2t_opt_prepare(const int32_t* day_idx, int B, const int32_t* seg_day, int nseg, int32_t* active,
                    void* stream);
int b2t_grad_norm_clip_f32(const float* grads, const int32_t* chunk2seg, const int32_t* active,
                           int nchunks, float max_norm, float* partial_ws, float* out3,
                           int32_t* seg_step, int nseg, void* stream);
/* AdamW (torch.optim.AdamW, rnn_trainer.py:283-290), active tensors only, k = seg_step (1-based):
 *   g *= clip3[2]; p *= 1-lr*wd; m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
 *   p -= lr/(1-b1^k) * m / (sqrt(v)/sqrt(1-b2^k) + eps)      (bias corrections evaluated in fp64)
 * lr3_host / wd3_host: HOST arrays of 3 floats (per group: bias, day, other).  clip3 may be NULL. */
int b2t_adamw_f32(float* params, float* grads, float* exp_avg, float* exp_avg_sq,
                  const int32_t* chunk2seg, const int32_t* active, const int32_t* seg_group,
                  const int32_t* seg_step, int nchunks, const float* clip3, const float* lr3_host,
                  const float* wd3_host, double beta1, double beta2, float eps, void* stream);

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

#ifdef __cplusplus
}
#endif
#endif /* B2T_H */
