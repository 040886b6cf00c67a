// Shared device helpers for the fastmax HIP kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "../../include/fastmax_hip.h"

namespace fastmax {

// ---- element types ---------------------------------------------------------------
struct bf16_t { uint16_t bits; };
struct f16_t { _Float16 v; };

__device__ __forceinline__ float to_float(float x) { return x; }
__device__ __forceinline__ float to_float(bf16_t x) { return __uint_as_float(((uint32_t)x.bits) << 16); }
__device__ __forceinline__ float to_float(f16_t x) { return (float)x.v; }

__device__ __forceinline__ uint16_t f32_to_bf16_bits(float f) {
    // round-to-nearest-even; a plain cast lowers to v_cvt_pk_bf16_f32 on gfx950 (NaN stays NaN)
    __hip_bfloat16 h = __float2bfloat16(f);
    return *reinterpret_cast<uint16_t*>(&h);
}
template <typename T> __device__ __forceinline__ T from_float(float f);
template <> __device__ __forceinline__ float from_float<float>(float f) { return f; }
template <> __device__ __forceinline__ bf16_t from_float<bf16_t>(float f) { return bf16_t{f32_to_bf16_bits(f)}; }
template <> __device__ __forceinline__ f16_t from_float<f16_t>(float f) { return f16_t{(_Float16)f}; }

// ---- strided (B,H,N,D) view ------------------------------------------------------
struct View {
    const void* ptr;
    int64_t sb, sh, sn;   // element strides; D has stride 1
};
struct Strides3 { int64_t sb, sh, sn; };

template <typename T>
__device__ __forceinline__ const T* row_ptr(const void* base, int64_t sb, int64_t sh, int64_t sn, int b, int h,
                                            int n) {
    return reinterpret_cast<const T*>(base) + (int64_t)b * sb + (int64_t)h * sh + (int64_t)n * sn;
}

// ---- wave helpers (wave = 64 lanes) ------------------------------------------------
__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
    return x;
}
__device__ __forceinline__ float wave_max(float x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x = fmaxf(x, __shfl_xor(x, off, 64));
    return x;
}
// inclusive prefix sum across the 64 lanes of a wave (Hillis-Steele on __shfl_up)
__device__ __forceinline__ float wave_inclusive_scan(float x) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        float y = __shfl_up(x, off, 64);
        if (lane >= off) x += y;
    }
    return x;
}

template <int P> __device__ __forceinline__ float poly_f(float s) {
    if constexpr (P == 1) return 1.0f + s;
    else return 1.0f + s + 0.5f * s * s;
}
template <int P> __device__ __forceinline__ float poly_fprime(float s) {
    if constexpr (P == 1) return 1.0f;
    else return 1.0f + s;
}

}  // namespace fastmax

// ---- host-side launchers implemented in the .hip files ------------------------------
namespace fastmax {
// linearmax forward whose statistics are still to be computed (fastmax_hip_linearmax_forward_auto): the launcher fills inv_q /
// inv_k (B*H floats each) before its main kernel reads them; partials = scratch words (linearmax_stats_words(B, H, N))
struct LinearmaxStats {
    float *inv_q, *inv_k;
    unsigned int* partials;              // scratch: 32-bit words (paired statistics pass) or 64-bit (value, ~row) keys (state pass)
    int *nstar_q = nullptr, *nstar_k = nullptr;   // optional: the row that attains the max-norm, per head (-1: not determined)
};
struct FwdArgs {
    fastmax_problem prob;
    const void *q, *k, *v;
    Strides3 qs, ks, vs;
    void* o;
    float* g;
    void* workspace;
    size_t workspace_bytes;
    hipStream_t stream;
    const LinearmaxStats* stats = nullptr;
};
struct BwdArgs {
    fastmax_problem prob;
    const void *q, *k, *v, *o, *grad_o;
    const float* g;
    Strides3 qs, ks, vs, gos;
    void *dq, *dk, *dv;
    void* workspace;
    size_t workspace_bytes;
    hipStream_t stream;
    const float* fwd_states = nullptr;   // the forward's sequence-split prefix states, when the caller kept them
    // linearmax training route (fastmax_hip_linearmax_backward): q, k are the RAW tensors; the scan kernels apply the prologue
    // (x - mean) * scale[bh] while staging, as the fused forward does, and return the gradients wrt the normalised q, k
    const float *qscale = nullptr, *kscale = nullptr;
    int fuse_prologue = 0;               // bit 0: dk leaves as the gradient wrt the raw k (prologue backward inside the dK/dV kernel)
    const int* k_nstar = nullptr;        //        the row of k that attains the max-norm, per head (from the forward)
    const int* q_nstar = nullptr;        // bit 1: the same for dq / q (needs bit 0: the dK/dV kernel leaves the shared sum)
};

int launch_fwd_quadratic(const FwdArgs& a);
int launch_fwd_recurrent_p1(const FwdArgs& a);
int launch_fwd_mfma_p1(const FwdArgs& a);
int launch_fwd_mfma_p1_v2(const FwdArgs& a, int ablation);
bool mfma_p1_v2_supported(const FwdArgs& a);
// Run-time tuning knobs (A/B runs, ablations): read from the FASTMAX_* environment ONCE at first use, afterwards changed
// only through fastmax_hip_tune(); launch paths read a plain int, never getenv.
enum TuneKey { TUNE_MFMA_VARIANT = 0, TUNE_BF16_KERNEL = 1, TUNE_GEMM_SCHED = 2, TUNE_GEMM_GROUP_M = 3, TUNE_GEMM_XCD = 4, TUNE_COUNT };
int tune_get(int key);
int tune_set(const char* name, int value);
int tune_get_by_name(const char* name);
bool mfma_p1_supported(const fastmax_problem& p);
size_t mfma_p1_workspace(const fastmax_problem& p);
int launch_fwd_mfma_gen(const FwdArgs& a, const float* qscale, const float* kscale);
int launch_fwd_mfma_bf16(const FwdArgs& a, const float* qscale, const float* kscale);
bool mfma_bf16_supported(const fastmax_problem& p);
bool mfma_gen_supported(const fastmax_problem& p, bool norm);
int launch_normalize_cast(const void* x, Strides3 xs, int dtype, void* y, const float* inv_norm, int B, int H, int N, int D,
                          hipStream_t stream, const unsigned int* partials = nullptr, int npart = 0, float* inv_out = nullptr, int rep = 1);
int launch_normalize_partial_max(const void* x, Strides3 xs, int dtype, unsigned int* partials, int B, int H, int N, int D,
                                 hipStream_t stream);
size_t normalize_backward_workspace(int B, int H, int N);
size_t normalize_backward_workspace_grouped(int B, int G, int rep, int N);
int normalize_block_tokens(int rep);
int launch_normalize_backward(const void* x, Strides3 xs, int dtype, const void* gy, const float* inv_norm, void* gx, int B, int H,
                              int N, int D, void* ws, hipStream_t stream, int rep = 1);
int launch_normalize_stats(const void* x, Strides3 xs, int dtype, float* inv_norm, int B, int H, int N, int D,
                           void* workspace, hipStream_t stream);
int launch_normalize_stats2(const void* x0, Strides3 s0, const void* x1, Strides3 s1, int dtype, float* inv0, float* inv1, int B, int H,
                            int N, int D, void* ws, hipStream_t stream, int* nstar0 = nullptr, int* nstar1 = nullptr);
// sequence split of the linear-time kernels when B*H alone cannot fill the chip
struct SplitPlan { int nseg, cps; };
SplitPlan split_plan(const fastmax_problem& p);
size_t split_workspace_bytes(const fastmax_problem& p, int dp);
int launch_split_states(const FwdArgs& a, const SplitPlan& plan, int dp, const float* kscale);
int linearmax_stats_and_states(const FwdArgs& a, const SplitPlan& plan, int dp);
bool unmasked_lin_supported(const fastmax_problem& p);
size_t unmasked_lin_workspace(const fastmax_problem& p);
int launch_fwd_unmasked_p1(const FwdArgs& a);
int launch_bwd_unmasked_p1(const BwdArgs& a);
size_t unmasked_lin_bwd_workspace(const fastmax_problem& p);
bool unmasked_lin_bwd_supported(const fastmax_problem& p);
int launch_bwd_prep_c(const BwdArgs& a, float* cbuf);
int launch_normalize_fixadd(const void* x, Strides3 xs, int dtype, const float* inv_norm, const float* part_dot,
                            const int* nstar, int nblk, int B, int H, int N, int D, void* gx, hipStream_t stream);
int launch_fwd_mfma_d128_2p(const FwdArgs& a, const float* qscale, const float* kscale);
bool mfma_d128_2p_supported(const fastmax_problem& p);
int launch_split_rstates(const void* q, Strides3 qs, const void* go, Strides3 gos, const float* g, const float* c, float* state,
                         const fastmax_problem& p, const SplitPlan& plan, int dp, hipStream_t stream, const float* qscale = nullptr,
                         int first_seg = 1);
size_t lin_bwd_workspace(const fastmax_problem& p);
int launch_fwd_quad_mfma(const FwdArgs& a);
bool quad_mfma_supported(const fastmax_problem& p);
int launch_fwd_quad32(const FwdArgs& a);
bool quad32_supported(const fastmax_problem& p);
int launch_bwd_quadratic(const BwdArgs& a);
int launch_bwd_quad_mfma(const BwdArgs& a);
bool quad_mfma_bwd_supported(const fastmax_problem& p);
int launch_bwd_quad32(const BwdArgs& a);
int launch_bwd_quad32_main(const BwdArgs& a);
bool quad32_bwd_supported(const fastmax_problem& p);
bool quad32_bwd_layout_ok(const BwdArgs& a);
size_t quad32_bwd_gt_offset(const fastmax_problem& p);
int launch_bwd_lin(const BwdArgs& a);
int launch_bwd_scan(const BwdArgs& a);
bool scan_bwd_supported(const fastmax_problem& p);
size_t scan_bwd_workspace(const fastmax_problem& p);
bool lin_bwd_supported(const fastmax_problem& p);
size_t bwd_quadratic_workspace(const fastmax_problem& p);
int launch_normalize(const void* x, Strides3 xs, int dtype, float* y, float* inv_norm, int B, int H, int N, int D,
                     void* workspace, hipStream_t stream);
}  // namespace fastmax
