/* fastmax_hip.h -- C ABI of libfastmax_hip.so (MI355X / gfx950 only).
 *
 * This is the drop-in boundary for the reference's fastmax / linearmax operator.  The
 * reference is pure Python: there is no pre-existing C/FFI interface for this path (the
 * `fastmax_cuda.forwardpass/backwardpass` extension declared at setup_fast_cuda.py:24-34
 * and called at lit_gpt/model.py:82-84,116 has no sources in the repo and is a different,
 * RPE-bearing op).  Each entry point therefore cites the PYTHON function whose arithmetic
 * it replaces; the Python shim that binds them (ctypes) lives in
 * fastmax_experiments_amd/_lib.py and keeps the reference's call signatures.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is DEVICE memory unless stated
 *   - no allocation, no host synchronisation inside: the caller owns all buffers and
 *     the call is stream-ordered on `stream` (a hipStream_t passed as void*)
 *   - tensors are (B,H,N,D); `*_strides` are ELEMENT strides of the b, h, n dims
 *     (3 x int64, host memory); the D dim must have stride 1 and each row must start
 *     16-byte aligned (the Python shim copies anything else to contiguous first)
 *   - outputs are contiguous (B,H,N,D) / (B,H,N)
 *   - return 0 on success, a negative FASTMAX_E_* code on a rejected call, or a positive
 *     hipError_t if a launch failed
 */
#ifndef FASTMAX_HIP_H
#define FASTMAX_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FASTMAX_ABI_VERSION 8   /* 8: the lmhead_ce_* and debug_gemm_stamps entry points are gone: the head's two products are plain library GEMMs by decision, see DESIGN.md, and the losing GEMM loop variants were removed; + fastmax_hip_tune_get, fastmax_hip_build_flags, fastmax_hip_normalize_stats2(_workspace), fastmax_hip_lora_{down,tn,up}_dropout, fastmax_hip_lora_dropout_mask, fastmax_hip_linearmax_forward_auto(_workspace), fastmax_hip_linearmax_backward, fastmax_hip_linearmax_train_supported; 7: + qlora_gemm_rope; 6: + qlora_gemm, nf4_dequantize_transposed, lmhead_ce_*; 5: + nf4 *_s entry points (double-quantised block scales); 4: + fastmax_hip_tune; 2: + normalize_cast/backward, rope_qkv_split(_backward), cross_entropy_forward/backward; 3: + lora_down/tn/up/scatter, normalize_*_expand, forward_state_bytes, backward_with_states */

enum fastmax_dtype { FASTMAX_F32 = 0, FASTMAX_BF16 = 1, FASTMAX_F16 = 2 };

enum fastmax_error {
    FASTMAX_OK = 0,
    FASTMAX_E_BAD_P = -1,        /* p not in {1,2}: the Python shim raises ValueError like fastmax.py:362,428 */
    FASTMAX_E_BAD_SHAPE = -2,    /* non-positive size, causal with Nq != Nk, D > FASTMAX_MAX_D, B*H > 65535 */
    FASTMAX_E_BAD_DTYPE = -3,
    FASTMAX_E_WORKSPACE = -4,    /* workspace missing or too small */
    FASTMAX_E_ALIGNMENT = -5,    /* a base pointer or a stride breaks the 16-byte row alignment rule */
    FASTMAX_E_NULL = -6
};

#define FASTMAX_MAX_D 256   /* 136 .. 256: tile kernels only (matrix cores: forward every dtype, backward bf16; else vector ALU) */

/* Which kernel family a call would use; for tests and the benchmark report. */
enum fastmax_path {
    FASTMAX_PATH_AUTO = 0,
    FASTMAX_PATH_QUADRATIC = 1,  /* masked/unmasked f(QK^T)V tiles on the vector ALU, any p, any D<=128 */
    FASTMAX_PATH_RECURRENT = 2,  /* p=1 masked, carried K^T V' state, vector ALU, linear in N */
    FASTMAX_PATH_MFMA = 3,       /* p=1 masked, chunked scan on the matrix cores (split-bf16), linear in N */
    FASTMAX_PATH_QUADRATIC_MFMA = 4 /* f(QK^T)V tiles on the matrix cores: p=2, unmasked, N_q != N_k */
};

/* Run-time tuning knob for A/B runs and ablations (no reference counterpart).  The FASTMAX_* environment variables are read
 * once, at the first call into the library; afterwards a knob changes only through this call.  Keys: "mfma_variant" (headline
 * forward kernel generation / schedule), "bf16_kernel" (1 = all-MFMA bf16 scan, 0 = generic).  Host-only, not stream-ordered:
 * call it between launches.  Returns 0, or FASTMAX_E_BAD_SHAPE for an unknown key. */
int fastmax_hip_tune(const char* name, int value);
/* Current value of a tuning key (so that a benchmark line can record the state it was measured in); INT_MIN for an unknown
 * key.  fastmax_hip_build_flags(): bit 0 = the library was built with -DFASTMAX_ABLATIONS, i.e. it contains the timing-only
 * kernel variants whose RESULTS ARE WRONG (matrix instructions or memory passes removed); a production build has none and
 * fastmax_hip_tune refuses their "mfma_variant" numbers with FASTMAX_E_BAD_SHAPE. */
int fastmax_hip_tune_get(const char* name);
int fastmax_hip_build_flags(void);

typedef struct fastmax_problem {
    int B, H, Nq, Nk, D;
    int in_dtype;    /* enum fastmax_dtype of q,k,v (and grad_o)                       */
    int out_dtype;   /* enum fastmax_dtype of o; see dtype rule Q1 (grads use in_dtype) */
    int p;           /* 1 or 2: degree of the Taylor polynomial f                       */
    int causal;      /* 1 = `mask=True` of the reference (j <= i), needs Nq == Nk       */
    float a;         /* 1/nt          (nt: fastmax.py:78-82)                            */
    float b;         /* 1/(2 nt^2)    (used when p == 2)                                */
    float g0;        /* constant term of the denominator when causal == 0:
                        Nq for fastmax.py:269-271, Nk for fastmax_hack.py:21            */
    int path;        /* enum fastmax_path; FASTMAX_PATH_AUTO lets the library choose    */
} fastmax_problem;

/* ---- forward: replaces fastattention_einops.forward's F/g computation
 *      (attention_mechanisms/fastmax.py:84-97; compute_F_* 184-250, compute_g_* 252-322).
 *      o: (B,H,Nq,D) out_dtype.  g: (B,H,Nq) float32 denominator, saved for backward
 *      (fastmax.py:106); may be NULL when the caller does not need it.                   */
size_t fastmax_hip_forward_workspace(const fastmax_problem* prob);
int fastmax_hip_forward(const fastmax_problem* prob,
                        const void* q, const int64_t* q_strides,
                        const void* k, const int64_t* k_strides,
                        const void* v, const int64_t* v_strides,
                        void* o, float* g,
                        void* workspace, size_t workspace_bytes, void* stream);

/* ---- backward: replaces fastattention_einops.backward (fastmax.py:113-182) and the six
 *      gradient_o_{q,k,v}_{masked,unmasked} (383-691).  o, g: the forward's outputs.
 *      o: contiguous (B,H,Nq,D) in out_dtype.  grad_o: (B,H,Nq,D) in_dtype, strides as given.
 *      dq:(B,H,Nq,D) dk,dv:(B,H,Nk,D) contiguous, in_dtype (autograd hands each gradient
 *      back in the dtype of the input it belongs to).                                    */
size_t fastmax_hip_backward_workspace(const fastmax_problem* prob);
int fastmax_hip_backward(const fastmax_problem* prob,
                         const void* q, const int64_t* q_strides,
                         const void* k, const int64_t* k_strides,
                         const void* v, const int64_t* v_strides,
                         const void* o, const float* g,
                         const void* grad_o, const int64_t* go_strides,
                         void* dq, void* dk, void* dv,
                         void* workspace, size_t workspace_bytes, void* stream);
/*      The p=1 masked forward splits long sequences into segments and leaves the per-segment prefix states
 *      (sum k v^T, sum k) at the start of its workspace; the backward needs the same records.  A caller that keeps
 *      the forward's workspace alive hands it back here and the backward skips recomputing them:
 *      forward_state_bytes = how many leading bytes of the forward workspace hold them for this call (0: this
 *      problem / layout takes a kernel without a split -- pass NULL).                                            */
size_t fastmax_hip_forward_state_bytes(const fastmax_problem* prob,
                                       const void* q, const int64_t* q_strides,
                                       const void* k, const int64_t* k_strides,
                                       const void* v, const int64_t* v_strides, const void* o);
int fastmax_hip_backward_with_states(const fastmax_problem* prob,
                                     const void* q, const int64_t* q_strides,
                                     const void* k, const int64_t* k_strides,
                                     const void* v, const int64_t* v_strides,
                                     const void* o, const float* g,
                                     const void* grad_o, const int64_t* go_strides,
                                     void* dq, void* dk, void* dv,
                                     void* workspace, size_t workspace_bytes,
                                     const void* fwd_states, size_t fwd_state_bytes, void* stream);

/* ---- linearmax prologue: replaces fastattention_einops.normalize (fastmax.py:326-334)
 *      == the inline copy at fastmax_hack.py:38-43 / 10-15: per token subtract the mean
 *      over D, then divide the (b,h) slab by the max over tokens of the per-token L2 norm.
 *      x: (B,H,N,D) in `dtype`, strides given; y: contiguous (B,H,N,D) float32;
 *      inv_norm: (B,H) float32 = 1/max-norm (kept for the backward of the prologue).
 *      workspace: B*H floats.                                                            */
size_t fastmax_hip_normalize_workspace(int B, int H);
int fastmax_hip_normalize(const void* x, const int64_t* x_strides, int dtype,
                          float* y, float* inv_norm, int B, int H, int N, int D,
                          void* workspace, size_t workspace_bytes, void* stream);
/*      statistics only: inv_norm[b,h] = 1 / max_n ||x_n - mean_D x_n||  (no normalised copy is written) */
int fastmax_hip_normalize_stats(const void* x, const int64_t* x_strides, int dtype, float* inv_norm,
                                int B, int H, int N, int D, void* workspace, size_t workspace_bytes,
                                void* stream);
/* The same statistic for the two tensors the linearmax forward normalises (q and k, same shape and dtype), in two launches
 * in all: per-block maxima of both tensors, then one fold into inv_norm0 / inv_norm1 (B*H floats each).
 * workspace: fastmax_hip_normalize_stats2_workspace(B, H, N) bytes.  Reference: fastmax_hack.py:38-43. */
size_t fastmax_hip_normalize_stats2_workspace(int B, int H, int N);
int fastmax_hip_normalize_stats2(const void* x0, const int64_t* x0_strides, const void* x1, const int64_t* x1_strides, int dtype,
                                 float* inv_norm0, float* inv_norm1, int B, int H, int N, int D, void* workspace,
                                 size_t workspace_bytes, void* stream);

/*      training route: y = (x - mean_D x) / max-norm written in the INPUT dtype (contiguous (B,H,N,D)) -- the reference
 *      keeps 16-bit tensors 16-bit between the prologue and the attention (fastmax_hack.py:38-43) -- plus inv_norm (B,H).
 *      workspace: fastmax_hip_normalize_workspace(B, H); with 4 B H ceil(N/256) bytes or more the statistics pass writes one
 *      word per 256-token block and the row pass combines them (two launches instead of four, same values).
 *      FASTMAX_E_BAD_SHAPE when D is not a multiple of 16 bytes of elements (the caller then uses fastmax_hip_normalize). */
int fastmax_hip_normalize_cast(const void* x, const int64_t* x_strides, int dtype, void* y, float* inv_norm,
                               int B, int H, int N, int D, void* workspace, size_t workspace_bytes, void* stream);
/*      backward of the prologue (the reference gets it from autograd over fastmax_hack.py:38-43):
 *      grad_x = d/dx of y given grad_y; grad_y, grad_x contiguous (B,H,N,D) in `dtype`; x and inv_norm as in the forward.
 *      The max-norm term goes to the first token that attains the maximum (torch.argmax).  Bitwise reproducible.  */
size_t fastmax_hip_normalize_backward_workspace(int B, int H, int N);
int fastmax_hip_normalize_backward(const void* x, const int64_t* x_strides, int dtype, const void* grad_y,
                                   const float* inv_norm, void* grad_x, int B, int H, int N, int D,
                                   void* workspace, size_t workspace_bytes, void* stream);
/*      grouped-query form (GQA, lit_gpt/model.py:404-411): x holds the G key heads; y / grad_y hold the G * rep query-head
 *      copies (head g * rep + j) the attention reads.  The expand is fused into the forward's store, the sum over a group's
 *      heads into the backward; statistics and gradient are computed once per key head.  workspace: at least
 *      4 B G ceil(N/256) bytes (forward) / fastmax_hip_normalize_backward_workspace(B, G rep, N) (backward).                 */
int fastmax_hip_normalize_cast_expand(const void* x, const int64_t* x_strides, int dtype, void* y, float* inv_norm,
                                      int B, int G, int rep, int N, int D, void* workspace, size_t workspace_bytes,
                                      void* stream);
int fastmax_hip_normalize_backward_expand(const void* x, const int64_t* x_strides, int dtype, const void* grad_y,
                                          const float* inv_norm, void* grad_x, int B, int G, int rep, int N, int D,
                                          void* workspace, size_t workspace_bytes, void* stream);


/* ---- fused linearmax forward: fastmax_hack.py:36-60 (masked branch) in one pass over Q, K, V --
 *      the mean-centre / max-norm prologue is applied to the Q and K rows as they are staged, with the
 *      per-(b,h) scales from fastmax_hip_normalize_stats; then first-order masked fastmax with nt = 1
 *      (prob->a = 1, prob->p = 1, prob->causal = 1).  Returns FASTMAX_E_BAD_SHAPE when the shape is not
 *      covered by the matrix-core kernel (the caller then runs normalize + forward separately).     */
int fastmax_hip_linearmax_forward(const fastmax_problem* prob,
                                  const void* q, const int64_t* q_strides,
                                  const void* k, const int64_t* k_strides,
                                  const void* v, const int64_t* v_strides,
                                  const float* q_inv_norm, const float* k_inv_norm,
                                  void* o, float* g,
                                  void* workspace, size_t workspace_bytes, void* stream);
/*      workspace of the fused call = fastmax_hip_forward_workspace(prob) (sequence-split states)   */

/*      The same with the statistics computed by the call (the whole masked branch of fastmax_hack.py:36-60 in one entry point):
 *      q_inv_norm / k_inv_norm are OUTPUTS (B*H floats each).  When the sequence split is active the statistics ride on its state
 *      pass -- K is read there anyway and the state is linear in K's scale, so the prefix pass applies it; Q's statistic comes
 *      from extra blocks of the same launch -- otherwise they are fastmax_hip_normalize_stats2.                             */
size_t fastmax_hip_linearmax_forward_auto_workspace(const fastmax_problem* prob);
int fastmax_hip_linearmax_forward_auto(const fastmax_problem* prob,
                                       const void* q, const int64_t* q_strides,
                                       const void* k, const int64_t* k_strides,
                                       const void* v, const int64_t* v_strides,
                                       float* q_inv_norm, float* k_inv_norm,
                                       int* q_nstar, int* k_nstar,
                                       void* o, float* g,
                                       void* workspace, size_t workspace_bytes, void* stream);
/*      q_nstar / k_nstar (B*H ints each, or NULL): the row that attains the max-norm, per head (the statistics are kept as
 *      (value, row) keys); fastmax_hip_linearmax_backward's fused prologue backward needs them.                           */

/*      Training route of the same branch (masked, p = 1): the backward of fastmax_hip_linearmax_forward_auto.  q, k are the RAW
 *      tensors and q_inv_norm / k_inv_norm what the forward left; the linear-time scans apply the prologue while staging (as the
 *      forward does), so no normalised copy of q or k exists in memory.  dq, dk are the gradients wrt the NORMALISED q, k -- the
 *      caller finishes with fastmax_hip_normalize_backward(q, dq, q_inv_norm) and (k, dk, k_inv_norm) (the autograd the reference
 *      gets over fastmax_hack.py:38-43).  workspace = fastmax_hip_backward_workspace(prob); fwd_states = the forward's workspace
 *      (prefix states of the sequence split) or NULL.  _train_supported: 1 where both directions are covered (else the caller
 *      runs normalize_cast + fastmax_hip_forward / _backward).                                                              */
int fastmax_hip_linearmax_train_supported(const fastmax_problem* prob);
int fastmax_hip_linearmax_backward(const fastmax_problem* prob,
                                   const void* q, const int64_t* q_strides,
                                   const void* k, const int64_t* k_strides,
                                   const void* v, const int64_t* v_strides,
                                   const void* o, const float* g,
                                   const void* grad_o, const int64_t* go_strides,
                                   const float* q_inv_norm, const float* k_inv_norm,
                                   const int* q_nstar, const int* k_nstar,
                                   void* dq, void* dk, void* dv,
                                   void* workspace, size_t workspace_bytes,
                                   const void* fwd_states, size_t fwd_state_bytes, int flags, void* stream);
/*      flags bit 0 (needs k_nstar with every entry >= 0): dk leaves as the gradient wrt the RAW k (the dK/dV kernel applies
 *      inv (g - mean_D g) to its tile and a one-row fix-up adds the dL/dM term to row n*): no normalize_backward call for k.
 *      Only when every query head has its own k head.  flags bit 1 (needs q_nstar): the same for dq / q -- the number both
 *      fix-ups need, sum_ij dS_ij s_ij, is one and the same and comes out of the dK/dV kernel; bit 1 alone is the form for
 *      grouped-query heads (k a stride-0 view of the key heads: dk is summed over the group by the caller's prologue pass).  */


/* ---- the operator's neighbours in CausalSelfAttention.forward (lit_gpt/model.py:397-425) in one pass:
 *      qkv (B, T, G, q_per_kv + 2, head_size), the QKV linear's output  ->  q (B, G*q_per_kv, T, head_size),
 *      k, v (B, G*q_per_kv, T, head_size) when expand_kv != 0 (the reference's GQA expand) or (B, G, T, head_size);
 *      RoPE (apply_rope, model.py:702-708) on the first rope_n_elem elements of q and k with cos, sin: (T, rope_n_elem) float32.
 *      Replaces view/permute/split/expand/reshape copies + five elementwise launches + two cats per tensor.
 *      rope_n_elem / 2 and head_size - rope_n_elem must be multiples of 16 bytes of elements (else FASTMAX_E_BAD_SHAPE).
 *      expand_kv: bits 0-1 = 0 (k, v stay at G heads) / 1 (both repeated per query head) / 2 (only v);  bit 4
 *      (FASTMAX_ROPE_TABLES_16BIT): the caller's rope cache was in the tensors' own 16-bit dtype ("bf16-true"): the two products
 *      are rounded to that dtype before they are summed, which is what the tensor ops of model.py:708 then compute.
 *      _backward: gradients of q, k, v in the same layouts -> gradient of qkv (sum over the query heads of a group for k, v,
 *      inverse rotation, re-interleave).                                                                               */
#define FASTMAX_ROPE_TABLES_16BIT 16
int fastmax_hip_rope_qkv_split(const void* qkv, const float* cos, const float* sin, void* q, void* k, void* v,
                               int B, int T, int G, int q_per_kv, int head_size, int rope_n_elem, int expand_kv,
                               int dtype, void* stream);
int fastmax_hip_rope_qkv_split_backward(const void* grad_q, const void* grad_k, const void* grad_v, const float* cos,
                                        const float* sin, void* grad_qkv, int B, int T, int G, int q_per_kv,
                                        int head_size, int rope_n_elem, int expand_kv, int dtype, void* stream);

/* ---- loss path of the fine-tune step: the row-wise cross entropy behind chunked_cross_entropy
 *      (lit_gpt/utils.py:228-272; torch cross_entropy, reduction "none", ignore_index).
 *      logits: (M, V) in `dtype` with row stride ld (elements); targets: (M) int64.
 *      forward : lse[i] = log sum_v exp(z_iv);  loss[i] = lse[i] - z[i, t_i]   (0 where t_i == ignore_index)
 *      backward: grad_logits[i, v] = (softmax(z_i)[v] - [v == t_i]) * grad_loss[i] * grad_scale   (grad_loss may be NULL = 1;
 *                rows with an ignored target get 0); grad_logits may alias logits (in place).               */
int fastmax_hip_cross_entropy_forward(const void* logits, int64_t ld, const int64_t* targets, float* loss, float* lse,
                                      int64_t M, int V, int64_t ignore_index, int dtype, void* stream);
int fastmax_hip_cross_entropy_backward(const void* logits, int64_t ld, const int64_t* targets, const float* lse,
                                       const float* grad_loss, float grad_scale, void* grad_logits, int64_t ldg,
                                       int64_t M, int V, int64_t ignore_index, int dtype, void* stream);

/* ---- decode-time state cache (opt-in): O(D^2) per generated token instead of the reference's
 *      unmasked recompute over the zero-padded KV cache (lit_gpt/model.py:427-430,464-466, generate/base.py:85-92).
 *      state: per (b,h) record [S2 (DPxDP) | S1 (DP) | ksum (DP)] float32, DP = 64 (D <= 64) or 128.
 *      prefill_state: state of a whole prompt (k, v: (B,H,N,D)); prob->p = 1, prob->causal = 1.
 *      decode_step:   q,k,v of ONE new token ((B,H,1,D)); updates the state in place and writes
 *                     o (B,H,1,D) = masked first-order fastmax at the new last position; count_after = number of
 *                     tokens in the sequence including this one.  Not the reference's decode arithmetic (quirk Q4). */
size_t fastmax_hip_decode_state_bytes(int B, int H, int D);
int fastmax_hip_p1_prefill_state(const fastmax_problem* prob, const void* k, const int64_t* k_strides,
                                 const void* v, const int64_t* v_strides, float* state, void* stream);
int fastmax_hip_p1_decode_step(const void* q, const int64_t* q_strides, const void* k, const int64_t* k_strides,
                               const void* v, const int64_t* v_strides, float* state, void* o,
                               int B, int H, int D, int in_dtype, int out_dtype, float a, int64_t count_after,
                               void* stream);

/* ---- QLoRA linear: frozen NF4 base weight + LoRA branch, fused (csrc/nf4_lora.hip).
 *      Replaces the bitsandbytes Linear4bit matmul + the low-rank branch of
 *      lit_gpt/lora.py:170-177 (LoRALinear.forward) and :398-433 (LoRAQKVLinear.forward):
 *        y[M][N] = x[M][K] . deq(W)[N][K]^T + bias[N] + ea[M][32] . eb[N][32]^T
 *      wq: packed NF4 codes of the row-major (N,K) weight, two per byte, high nibble first;
 *      absmax: float32, one per 64 consecutive weights (block size 64); bias: float32 or NULL.
 *      ea = dropout(x) A^T and eb = scaling * scatter(lora_B) are bf16, rank padded to 32 columns
 *      (both NULL = no LoRA branch).  x, y: `dtype` (FASTMAX_BF16 or FASTMAX_F32), leading
 *      dimensions in elements.  Needs K % 64 == 0, N % 4 == 0, 16-byte aligned rows.
 *      NF4 follows the public QLoRA definition; parity with bitsandbytes is unpinned.           */
int fastmax_hip_nf4_linear_forward(const void* x, int64_t ldx, const uint8_t* wq, const float* absmax,
                                   const float* bias, const void* ea, const void* eb, void* y, int64_t ldy,
                                   int M, int N, int K, int dtype, void* stream);
/*      dx[M][K] = dy[M][N] . deq(W)[N][K]  (gradient wrt the input of the frozen base layer).
 *      Needs K % 128 == 0, N % 64 == 0.                                                         */
int fastmax_hip_nf4_linear_backward_input(const void* dy, int64_t lddy, const uint8_t* wq, const float* absmax,
                                          void* dx, int64_t lddx, int M, int N, int K, int dtype, void* stream);
/*      dense dequantisation (merge path: lora.py:142-168 dequantize + add LoRA + requantize)     */
int fastmax_hip_nf4_dequantize(const uint8_t* wq, const float* absmax, void* out, int64_t n, int dtype,
                               void* stream);

/* ---- double quantisation ("bnb.nf4-dq", finetune/lora.py:38; QLoRA arXiv 2305.14314 section 3): the fp32 block scales are
 *      themselves stored in 8 bits.  absmax[i] = code2[absmax_q[i]] * absmax2[i / 256] + offset.  The `_s` forms of the three
 *      entry points above take the scales as this struct (plain NF4: `absmax` set, `absmax_q` NULL).  The 256-entry map and
 *      the codec live on the host side (lora.py here); bitsandbytes is absent from the reference tree: parity UNPINNED.    */
typedef struct fastmax_nf4_scales {
    const float* absmax;      /* plain NF4: fp32, one per 64 weights; ignored when absmax_q != NULL */
    const uint8_t* absmax_q;  /* nf4-dq: 8-bit code per 64 weights, or NULL                          */
    const float* absmax2;     /* nf4-dq: fp32 scale per 256 codes                                    */
    const float* code2;       /* nf4-dq: the 256-entry map (device memory)                           */
    float offset;             /* nf4-dq: mean of the original absmax vector                          */
} fastmax_nf4_scales;
int fastmax_hip_nf4_linear_forward_s(const void* x, int64_t ldx, const uint8_t* wq, const fastmax_nf4_scales* scales,
                                     const float* bias, const void* ea, const void* eb, void* y, int64_t ldy,
                                     int M, int N, int K, int dtype, void* stream);
int fastmax_hip_nf4_linear_backward_input_s(const void* dy, int64_t lddy, const uint8_t* wq,
                                            const fastmax_nf4_scales* scales, void* dx, int64_t lddx, int M, int N, int K,
                                            int dtype, void* stream);
int fastmax_hip_nf4_dequantize_s(const uint8_t* wq, const fastmax_nf4_scales* scales, void* out, int64_t n, int dtype,
                                 void* stream);

/* ---- QLoRA linear at training row counts, one kernel (csrc/nf4_gemm.hip): 256 x 256 output tiles, x by LDS-DMA, the frozen
 *      weight decoded from NF4 in the loop (w_is_nf4 != 0, `scales` as above) or read as a dense bf16 (N, K) matrix
 *      (w_is_nf4 == 0: a LoRALinear on a dense base, lit_gpt/lora.py:170-177), the LoRA branch as one more 32-deep step:
 *        y[M][N] = x[M][K] . W[N][K]^T + bias[N] + ea[M][rank_pad] . eb[N][rank_pad]^T       (bf16 in / out, fp32 accumulate)
 *      Needs K % 64 == 0, N % 8 == 0, 16-byte aligned rows; rank_pad 16 or 32; bias float32 or NULL; ea, eb both or neither. */
int fastmax_hip_qlora_gemm(const void* x, int64_t ldx, const void* w, int w_is_nf4, const fastmax_nf4_scales* scales,
                           const float* bias, const void* ea, const void* eb, int rank_pad, void* y, int64_t ldy,
                           int M, int N, int K, void* stream);

/*      The qkv projection with its neighbours fused into the tile's way out: y = x W^T + bias + ea eb^T is never stored as
 *      (tokens, qkv features); the finished 256 x 256 tile is written as q (B, G q_per_kv, T, hs) and k, v (B, G, T, hs) with the
 *      rotation of lit_gpt/model.py:702-708 on the first rope_n_elem elements of the q and k heads -- lit_gpt/lora.py:419-433
 *      followed by model.py:397-425 in one kernel, bit-identical to fastmax_hip_qlora_gemm + fastmax_hip_rope_qkv_split.
 *      x rows are b * T + t; dense bf16 weight [N][K]; N == G (q_per_kv + 2) head_size; 256 % head_size == 0;
 *      rope_n_elem % 16 == 0; K % 64 == 0; cos, sin (T, rope_n_elem) float32; tables16 as in fastmax_hip_rope_qkv_split.     */
int fastmax_hip_qlora_gemm_rope(const void* x, int64_t ldx, const void* w, const float* bias, const void* ea, const void* eb,
                                int rank_pad, const float* cos, const float* sin, void* q, void* k, void* v, int M, int N, int K,
                                int T, int G, int q_per_kv, int head_size, int rope_n_elem, int tables16, void* stream);

/*      W^T as dense bf16 [K][N] from the codes of W [N][K]: the weight operand of dx = dy . W through fastmax_hip_qlora_gemm
 *      (x := dy, w := W^T, M x K output).  N % 64 == 0, K % 64 == 0.                                                       */
int fastmax_hip_nf4_dequantize_transposed(const uint8_t* wq, const fastmax_nf4_scales* scales, void* out, int N, int K,
                                          void* stream);

/* ---- QLoRA linear at training sizes: the rank-r products around the library GEMM of the frozen weight
 *      (csrc/lora_thin.hip).  Replaces the tensor ops of lit_gpt/lora.py:170-177 / :419-433 and their autograd mirror
 *      when the base product runs as a dense GEMM.  All matrices bf16 row-major, leading dimensions in elements,
 *      fp32 accumulation; RP = rank padded with zero columns to 16 or 32.
 *        down: e[M][RP] = x[M][K] . bt[RP][K]^T;  et (or NULL) receives e^T as [RP][ldet], zero from column M to ldet
 *              (ldet a multiple of 16, >= M).  K % 64 == 0, 16-byte aligned rows.                                  */
int fastmax_hip_lora_down(const void* x, int64_t ldx, const void* bt, int64_t ldbt, void* e, int64_t lde, void* et,
                          int64_t ldet, int M, int K, int RP, void* stream);
/*        tn:   out = et[R][M] . x[M][ncols], written as [R][ncols] (or [ncols][R] when `transpose`) in out_dtype
 *              (FASTMAX_F32 / FASTMAX_BF16), fp32 sums in a fixed order;  et as written by lora_down (zero padded to a
 *              multiple of 16 columns), R <= RP valid rows.  ncols % 64 == 0.  workspace: fastmax_hip_lora_tn_workspace bytes. */
int64_t fastmax_hip_lora_tn_workspace(int M, int ncols, int RP);
int fastmax_hip_lora_tn(const void* et, int64_t ldet, const void* x, int64_t ldx, void* out, int out_dtype, int transpose,
                        int R, void* workspace, int M, int ncols, int RP, void* stream);
/*        up:   y[M][N] += e[M][R] . bn[N][R]^T (+ bias[N], float32 or NULL), in place; bn as (N, R) rows, or as (R, N)
 *              rows when bn_transposed.  R in {8,16,24,32}, N % 8 == 0.                                                */
int fastmax_hip_lora_up(void* y, int64_t ldy, const void* e, int64_t lde, const void* bn, int64_t ldb, int bn_transposed,
                        const float* bias, int M, int N, int R, void* stream);
/*        LoRA dropout (lit_gpt/lora.py:175, 422 `self.lora_dropout(x)`; finetune/lora.py:42 lora_dropout = 0.05) inside the
 *        same three kernels.  The keep mask of the branch input x (M, K) is a counter-based function of (seed[0], row,
 *        column) -- 16 hash bits per element, kept iff >= round(65536 p_drop) -- regenerated wherever it is needed and never
 *        stored; `seed` is a DEVICE pointer (one 32-bit word is read), so a captured HIP graph can draw a new mask per replay.
 *          down_dropout: e = dropout(x) . bt^T            (forward: x A^T)
 *          tn_dropout:   out = et . dropout(x)            (backward: dA = d_ea^T dropout(x); x in the X role)
 *          up_dropout:   y += mask o (e . bn^T) / (1 - p) (backward: dx of the branch; y has the shape of x)
 *        with dropout(x) = mask o x / (1 - p); seed == NULL or p_drop <= 0 gives the plain product.
 *        dropout_mask writes the M x K mask as bytes (1 = kept), for tests and inspection.                              */
int fastmax_hip_lora_down_dropout(const void* x, int64_t ldx, const void* bt, int64_t ldbt, void* e, int64_t lde, void* et,
                                  int64_t ldet, int M, int K, int RP, const void* seed, float p_drop, void* stream);
int fastmax_hip_lora_tn_dropout(const void* et, int64_t ldet, const void* x, int64_t ldx, void* out, int out_dtype, int transpose,
                                int R, void* workspace, int M, int ncols, int RP, const void* seed, float p_drop, void* stream);
int fastmax_hip_lora_up_dropout(void* y, int64_t ldy, const void* e, int64_t lde, const void* bn, int64_t ldb, int bn_transposed,
                                const float* bias, int M, int N, int R, const void* seed, float p_drop, void* stream);
int fastmax_hip_lora_dropout_mask(void* mask, int M, int K, const void* seed, float p_drop, void* stream);
/*        scatter: the (RP, N) bf16 operand of the branch from lora_B (n_rows, r; b_dtype F32 / BF16):
 *              et[part r + j][n] = scaling b[rowmap[part][n]][j], 0 where rowmap is -1 and in rows >= n_parts r
 *              (LoRAQKVLinear's lora_ind / zero_pad, lit_gpt/lora.py:263-342; one part with the identity map = LoRALinear).
 *              backward: db[i][j] = scaling d_et[part[i] r + j][ind[i]]  (d_dtype F32 / BF16).                          */
int fastmax_hip_lora_scatter(const void* b, int b_dtype, int r, const int32_t* rowmap, int n_parts, float scaling, void* et,
                             int64_t ldet, int N, int RP, void* stream);
int fastmax_hip_lora_scatter_backward(const void* d_et, int d_dtype, int64_t ldd, const int32_t* ind, const int32_t* part,
                                      float scaling, void* db, int b_dtype, int n_rows, int r, void* stream);

/* ---- introspection */
int fastmax_hip_abi_version(void);
/* path a problem would take under FASTMAX_PATH_AUTO (enum fastmax_path), <0 on a bad problem */
int fastmax_hip_select_path(const fastmax_problem* prob);
const char* fastmax_hip_error_string(int code);

#ifdef __cplusplus
}
#endif
#endif /* FASTMAX_HIP_H */
