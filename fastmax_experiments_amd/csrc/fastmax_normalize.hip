// linearmax prologue for the training route (reference: attention_mechanisms/fastmax_hack.py:38-43, fastmax.py:326-334):
//     y = (x - mean_D x) / M,   M = max_n || x_n - mean_D x_n ||      per (b,h)
// written in the INPUT dtype (the reference keeps 16-bit tensors 16-bit between the prologue and the attention), and its
// backward (the reference gets it from autograd over the same ops):
//     gxc = gy / M;   dL/dM = -sum_{n,d}(gy y) / M;   gxc[n*] += dL/dM y[n*]  (n* = first argmax_n of the row norm);
//     gx = gxc - mean_D gxc
// Rows are spread over LPR lanes with 16-byte loads (scalar loads for unaligned views).  The backward is two launches:
// the row pass (every row without the dL/dM term + per-block partials sum gy.xc, best (norm^2, n)), then one wave per head
// that combines the partials in a fixed order (bitwise reproducible) and rewrites the one row that has the extra term.
#include <stdlib.h>

#include "fastmax_common.h"

namespace fastmax {

typedef unsigned int nu32x4 __attribute__((ext_vector_type(4)));

template <typename T> struct RowPiece {
    static constexpr int EPL = 16 / sizeof(T);
    float v[EPL];
};

// elements [sub*EPL, sub*EPL + EPL) of a row as floats (zero past D)
template <typename T>
__device__ __forceinline__ void load_row_piece(const T* row, int sub, int D, int vec, float (&v)[16 / sizeof(T)]) {
    constexpr int EPL = 16 / sizeof(T);
    if (vec) {
        nu32x4 raw = {0, 0, 0, 0};
        if (sub * EPL < D) raw = *reinterpret_cast<const nu32x4*>(row + sub * EPL);
        const T* pv = reinterpret_cast<const T*>(&raw);
#pragma unroll
        for (int e = 0; e < EPL; ++e) v[e] = to_float(pv[e]);
    } else {
#pragma unroll
        for (int e = 0; e < EPL; ++e) v[e] = (sub * EPL + e) < D ? to_float(row[sub * EPL + e]) : 0.f;
    }
}
template <typename T>
__device__ __forceinline__ void store_row_piece(T* row, int sub, int D, int vec, const float (&v)[16 / sizeof(T)]) {
    constexpr int EPL = 16 / sizeof(T);
    if (vec) {
        if (sub * EPL < D) {
            nu32x4 raw;
            T* pv = reinterpret_cast<T*>(&raw);
#pragma unroll
            for (int e = 0; e < EPL; ++e) pv[e] = from_float<T>(v[e]);
            *reinterpret_cast<nu32x4*>(row + sub * EPL) = raw;
        }
    } else {
#pragma unroll
        for (int e = 0; e < EPL; ++e)
            if (sub * EPL + e < D) row[sub * EPL + e] = from_float<T>(v[e]);
    }
}
template <int LPR> __device__ __forceinline__ float group_sum(float s) {
#pragma unroll
    for (int off = 1; off < LPR; off <<= 1) s += __shfl_xor(s, off, 64);
    return s;
}

// ---- forward: y (dtype T, contiguous) = (x - mean) * inv[bh];  grid = (ceil(N / (256/LPR) / RPT), B*H) -----------------
template <typename T, int LPR>
__global__ __launch_bounds__(256) void normalize_cast_kernel(const void* x, Strides3 xs, int H, int N, int D, const float* inv_norm,
                                                             T* y, int vec, const unsigned int* partials, int npart, float* inv_out, int rep, int tok) {
    constexpr int EPL = 16 / sizeof(T), RPB = 256 / LPR;
    const int tid = threadIdx.x, sub = tid % LPR, rgrp = tid / LPR;
    const int bh = blockIdx.y, b = bh / H, h = bh % H;
    float inv;
    if (partials) {
        // the head's maximum from the per-block words of normalize_max_kernel (same fmaxf chain as the atomic form)
        float m = 0.f;
        for (int i = 0; i < npart; ++i) m = fmaxf(m, __uint_as_float(partials[(int64_t)bh * npart + i]));
        inv = 1.0f / sqrtf(m);
        if (blockIdx.x == 0 && tid == 0) inv_out[bh] = inv;
    } else {
        inv = inv_norm[bh];
    }
    const float invD = 1.0f / (float)D;
    const int n_begin = blockIdx.x * tok, n_end = min(N, n_begin + tok);
    for (int n = n_begin + rgrp; n < n_end; n += RPB) {
        float v[EPL];
        load_row_piece<T>(row_ptr<T>(x, xs.sb, xs.sh, xs.sn, b, h, n), sub, D, vec, v);
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) s += v[e];
        const float mean = group_sum<LPR>(s) * invD;
#pragma unroll
        for (int e = 0; e < EPL; ++e) v[e] = (v[e] - mean) * inv;
        // rep > 1: the row is written for the rep query heads that share this key head (GQA expand fused into the store)
        for (int j = 0; j < rep; ++j) store_row_piece<T>(y + (((int64_t)bh * rep + j) * N + n) * D, sub, D, vec, v);
    }
}

// ---- backward, pass 1: per block  sum_n gy_n . xc_n  and the best (||xc_n||^2, first n) ------------------------------------
template <typename T, int LPR>
__global__ __launch_bounds__(256) void normalize_bwd_reduce_kernel(const void* x, Strides3 xs, const T* gy, int H, int N, int D,
                                                                   float* part_dot, unsigned long long* part_best, int vec) {
    constexpr int EPL = 16 / sizeof(T), RPB = 256 / LPR, TOK = 256;
    __shared__ float sdot[4];
    __shared__ unsigned long long sbest[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, sub = tid % LPR, rgrp = tid / LPR;
    const int bh = blockIdx.y, b = bh / H, h = bh % H;
    const float invD = 1.0f / (float)D;
    const int n_begin = blockIdx.x * TOK, n_end = min(N, n_begin + TOK);
    float dot = 0.f;
    unsigned long long best = 0ull;
    for (int n = n_begin + rgrp; n < n_end; n += RPB) {
        float v[EPL], gv[EPL];
        load_row_piece<T>(row_ptr<T>(x, xs.sb, xs.sh, xs.sn, b, h, n), sub, D, vec, v);
        load_row_piece<T>(gy + ((int64_t)bh * N + n) * D, sub, D, vec, gv);
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) s += v[e];
        const float mean = group_sum<LPR>(s) * invD;
        float nn = 0.f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const float c = (sub * EPL + e) < D ? v[e] - mean : 0.f;
            nn = fmaf(c, c, nn);
            dot = fmaf(gv[e], c, dot);
        }
        nn = group_sum<LPR>(nn);
        // squared norms are >= 0: their bit patterns order like unsigned integers; ties -> the smallest n (torch.argmax)
        const unsigned long long key = ((unsigned long long)__float_as_uint(nn) << 32) | (unsigned long long)(0xffffffffu - (unsigned)n);
        best = key > best ? key : best;
    }
    dot = wave_sum(dot);
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long o = __shfl_xor(best, off, 64);
        best = o > best ? o : best;
    }
    if (lane == 0) { sdot[wave] = dot; sbest[wave] = best; }
    __syncthreads();
    if (tid == 0) {
        const int idx = bh * gridDim.x + blockIdx.x;
        part_dot[idx] = (sdot[0] + sdot[1]) + (sdot[2] + sdot[3]);
        unsigned long long m = sbest[0];
#pragma unroll
        for (int i = 1; i < 4; ++i) m = sbest[i] > m ? sbest[i] : m;
        part_best[idx] = m;
    }
}

// ---- backward, pass 2: the row pass ---------------------------------------------------------------------------------------
template <typename T, int LPR>
__global__ __launch_bounds__(256) void normalize_bwd_apply_kernel(const void* x, Strides3 xs, const T* gy, const float* inv_norm,
                                                                  const float* part_dot, const unsigned long long* part_best,
                                                                  int H, int N, int D, T* gx, int vec) {
    constexpr int EPL = 16 / sizeof(T), RPB = 256 / LPR, TOK = 256;
    const int tid = threadIdx.x, sub = tid % LPR, rgrp = tid / LPR;
    const int bh = blockIdx.y, b = bh / H, h = bh % H;
    const float inv = inv_norm[bh], invD = 1.0f / (float)D;
    float S = 0.f;
    unsigned long long best = 0ull;
    for (int i = 0; i < (int)gridDim.x; ++i) {                     // fixed order: reproducible
        S += part_dot[bh * gridDim.x + i];
        const unsigned long long o = part_best[bh * gridDim.x + i];
        best = o > best ? o : best;
    }
    const int nstar = (int)(0xffffffffu - (unsigned)(best & 0xffffffffull));
    const float dLdM = -(S * inv) * inv;                           // -sum(gy y) / M,  y = xc / M
    const int n_begin = blockIdx.x * TOK, n_end = min(N, n_begin + TOK);
    for (int n = n_begin + rgrp; n < n_end; n += RPB) {
        float gv[EPL];
        load_row_piece<T>(gy + ((int64_t)bh * N + n) * D, sub, D, vec, gv);
#pragma unroll
        for (int e = 0; e < EPL; ++e) gv[e] *= inv;
        if (n == nstar) {                                          // uniform per row group
            float v[EPL];
            load_row_piece<T>(row_ptr<T>(x, xs.sb, xs.sh, xs.sn, b, h, n), sub, D, vec, v);
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < EPL; ++e) s += v[e];
            const float mean = group_sum<LPR>(s) * invD;
#pragma unroll
            for (int e = 0; e < EPL; ++e)
                if (sub * EPL + e < D) gv[e] = fmaf(dLdM, (v[e] - mean) * inv, gv[e]);
        }
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) s += gv[e];
        const float gm = group_sum<LPR>(s) * invD;
#pragma unroll
        for (int e = 0; e < EPL; ++e) gv[e] -= gm;
        store_row_piece<T>(gx + ((int64_t)bh * N + n) * D, sub, D, vec, gv);
    }
}

// ---- backward in one row pass + a one-row fix-up ----------------------------------------------------------------------------
// Only the row that attains the maximum (n*) sees the dL/dM term, so every row can be written as  gx = inv gy - mean_D(inv gy)
// in the SAME pass that accumulates the block's  sum gy.xc  and best (||xc||^2, n); a one-workgroup-per-head kernel then
// combines the partials (fixed order) and rewrites row n* with the full formula.  x and gy are read once instead of twice;
// same formulas as the reduce + apply pair above (kept for A/B runs); results differ from it by last-bit float32 rounding only
// (the compiler contracts the two forms differently).
template <typename T, int LPR>
__global__ __launch_bounds__(256) void normalize_bwd_rows_kernel(const void* x, Strides3 xs, const T* gy, const float* inv_norm, int H, int N,
                                                                 int D, T* gx, float* part_dot, unsigned long long* part_best, int vec, int rep, int tok) {
    constexpr int EPL = 16 / sizeof(T), RPB = 256 / LPR;
    __shared__ float sdot[4];
    __shared__ unsigned long long sbest[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, sub = tid % LPR, rgrp = tid / LPR;
    const int bh = blockIdx.y, b = bh / H, h = bh % H;
    const float inv = inv_norm[bh], invD = 1.0f / (float)D;
    const int n_begin = blockIdx.x * tok, n_end = min(N, n_begin + tok);
    float dot = 0.f;
    unsigned long long best = 0ull;
    for (int n = n_begin + rgrp; n < n_end; n += RPB) {
        float v[EPL], gv[EPL];
        load_row_piece<T>(row_ptr<T>(x, xs.sb, xs.sh, xs.sn, b, h, n), sub, D, vec, v);
        load_row_piece<T>(gy + (((int64_t)bh * rep) * N + n) * D, sub, D, vec, gv);
        for (int j = 1; j < rep; ++j) {                            // y was written for rep heads: their gradients add up
            float gj[EPL];
            load_row_piece<T>(gy + (((int64_t)bh * rep + j) * N + n) * D, sub, D, vec, gj);
#pragma unroll
            for (int e = 0; e < EPL; ++e) gv[e] += gj[e];
        }
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) s += v[e];
        const float mean = group_sum<LPR>(s) * invD;
        float nn = 0.f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const float c = (sub * EPL + e) < D ? v[e] - mean : 0.f;
            nn = fmaf(c, c, nn);
            dot = fmaf(gv[e], c, dot);
        }
        nn = group_sum<LPR>(nn);
        const unsigned long long key = ((unsigned long long)__float_as_uint(nn) << 32) | (unsigned long long)(0xffffffffu - (unsigned)n);
        best = key > best ? key : best;
        // the row's gradient without the dL/dM term (exact for every row but n*)
#pragma unroll
        for (int e = 0; e < EPL; ++e) gv[e] *= inv;
        float gs = 0.f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) gs += gv[e];
        const float gm = group_sum<LPR>(gs) * invD;
#pragma unroll
        for (int e = 0; e < EPL; ++e) gv[e] -= gm;
        store_row_piece<T>(gx + ((int64_t)bh * N + n) * D, sub, D, vec, gv);
    }
    dot = wave_sum(dot);
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long o = __shfl_xor(best, off, 64);
        best = o > best ? o : best;
    }
    if (lane == 0) { sdot[wave] = dot; sbest[wave] = best; }
    __syncthreads();
    if (tid == 0) {
        const int idx = bh * gridDim.x + blockIdx.x;
        part_dot[idx] = (sdot[0] + sdot[1]) + (sdot[2] + sdot[3]);
        unsigned long long m = sbest[0];
#pragma unroll
        for (int i = 1; i < 4; ++i) m = sbest[i] > m ? sbest[i] : m;
        part_best[idx] = m;
    }
}

// one wave per head: combine the partials, rewrite row n* with the dL/dM term (the n == nstar branch of the apply kernel)
template <typename T, int LPR>
__global__ __launch_bounds__(64) void normalize_bwd_fix_kernel(const void* x, Strides3 xs, const T* gy, const float* inv_norm,
                                                               const float* part_dot, const unsigned long long* part_best, int nblk, int H,
                                                               int N, int D, T* gx, int vec, int rep) {
    constexpr int EPL = 16 / sizeof(T);
    const int tid = threadIdx.x, sub = tid % LPR;
    const int bh = blockIdx.x, b = bh / H, h = bh % H;
    const float inv = inv_norm[bh], invD = 1.0f / (float)D;
    // lane l takes partials l, l + 64, ...; the lanes are combined by a fixed butterfly: the same order every run
    float S = 0.f;
    unsigned long long best = 0ull;
    for (int i = tid; i < nblk; i += 64) {
        S += part_dot[(int64_t)bh * nblk + i];
        const unsigned long long o = part_best[(int64_t)bh * nblk + i];
        best = o > best ? o : best;
    }
    S = wave_sum(S);
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long o = __shfl_xor(best, off, 64);
        best = o > best ? o : best;
    }
    const int n = (int)(0xffffffffu - (unsigned)(best & 0xffffffffull));
    const float dLdM = -(S * inv) * inv;
    if (tid >= LPR || n < 0 || n >= N) return;
    float gv[EPL], v[EPL];
    load_row_piece<T>(gy + (((int64_t)bh * rep) * N + n) * D, sub, D, vec, gv);
    for (int j = 1; j < rep; ++j) {
        float gj[EPL];
        load_row_piece<T>(gy + (((int64_t)bh * rep + j) * N + n) * D, sub, D, vec, gj);
#pragma unroll
        for (int e = 0; e < EPL; ++e) gv[e] += gj[e];
    }
#pragma unroll
    for (int e = 0; e < EPL; ++e) gv[e] *= inv;
    load_row_piece<T>(row_ptr<T>(x, xs.sb, xs.sh, xs.sn, b, h, n), sub, D, vec, v);
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) s += v[e];
    const float mean = group_sum<LPR>(s) * invD;
#pragma unroll
    for (int e = 0; e < EPL; ++e)
        if (sub * EPL + e < D) gv[e] = fmaf(dLdM, (v[e] - mean) * inv, gv[e]);
    float gs = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) gs += gv[e];
    const float gm = group_sum<LPR>(gs) * invD;
#pragma unroll
    for (int e = 0; e < EPL; ++e) gv[e] -= gm;
    store_row_piece<T>(gx + ((int64_t)bh * N + n) * D, sub, D, vec, gv);
}

// The same fix-up for a gradient whose rows ALREADY carry inv (g - mean_D g) (written by the scan kernel that produced g, which also
// left the partial sums): the dL/dM term is added to row n* in place, gx[n*] += dLdM (x_n* - mean) inv.  One wave per head.
template <typename T, int LPR>
__global__ __launch_bounds__(64) void normalize_bwd_fixadd_kernel(const void* x, Strides3 xs, const float* inv_norm, const float* part_dot,
                                                                  const int* nstar, int nblk, int H, int N, int D, T* gx, int vec) {
    constexpr int EPL = 16 / sizeof(T);
    const int tid = threadIdx.x, sub = tid % LPR;
    const int bh = blockIdx.x, b = bh / H, h = bh % H;
    const float inv = inv_norm[bh], invD = 1.0f / (float)D;
    float S = 0.f;
    for (int i = tid; i < nblk; i += 64) S += part_dot[(int64_t)bh * nblk + i];
    S = wave_sum(S);
    const int n = nstar[bh];                                       // the row that attains the max-norm (the forward found it)
    // part_dot holds T = sum_n g_n . y_n (y = xc inv): dL/dM = -(T / inv) inv^2 = -T inv
    const float dLdM = -S * inv;
    if (tid >= LPR || n < 0 || n >= N) return;
    float v[EPL], gv[EPL];
    load_row_piece<T>(row_ptr<T>(x, xs.sb, xs.sh, xs.sn, b, h, n), sub, D, vec, v);
    load_row_piece<T>(gx + ((int64_t)bh * N + n) * D, sub, D, vec, gv);
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) s += v[e];
    const float mean = group_sum<LPR>(s) * invD;
#pragma unroll
    for (int e = 0; e < EPL; ++e)
        if (sub * EPL + e < D) gv[e] = fmaf(dLdM, (v[e] - mean) * inv, gv[e]);
    store_row_piece<T>(gx + ((int64_t)bh * N + n) * D, sub, D, vec, gv);
}

static int nrm_vec_ok(const void* x, Strides3 xs, size_t es, int D) {
    const int epl = (int)(16 / es);
    return (reinterpret_cast<uintptr_t>(x) % 16 == 0) && ((xs.sb * es) % 16 == 0) && ((xs.sh * es) % 16 == 0) &&
           ((xs.sn * es) % 16 == 0) && (D % epl == 0);
}
#define NRM_LPR_SWITCH(need, CALL)      \
    if ((need) <= 4) { CALL(4); }       \
    else if ((need) <= 8) { CALL(8); }  \
    else if ((need) <= 16) { CALL(16); } \
    else if ((need) <= 32) { CALL(32); } \
    else { CALL(64); }

template <typename T>
static int normalize_cast_t(const void* x, Strides3 xs, void* y, const float* inv_norm, int B, int H, int N, int D, hipStream_t stream,
                            const unsigned int* partials, int npart, float* inv_out, int rep) {
    const int epl = (int)(16 / sizeof(T)), need = (D + epl - 1) / epl;
    if (need > 64) return FASTMAX_E_BAD_SHAPE;
    // 16-byte accesses need whole pieces per row and aligned rows on both sides; otherwise element-wise loads / stores
    const int vec = nrm_vec_ok(x, xs, sizeof(T), D) && !(reinterpret_cast<uintptr_t>(y) & 15);
    // rep > 1: each row is stored rep times, so blocks take ~256 / rep tokens (at least 32) to keep the grid as large
    const int tok = normalize_block_tokens(rep);
    const dim3 grid((N + tok - 1) / tok, B * H), block(256);
#define CALL(L) hipLaunchKernelGGL((normalize_cast_kernel<T, L>), grid, block, 0, stream, x, xs, H, N, D, inv_norm, reinterpret_cast<T*>(y), vec, partials, npart, inv_out, rep, tok)
    NRM_LPR_SWITCH(need, CALL)
#undef CALL
    return (int)hipGetLastError();
}
// partials == nullptr: inv_norm is an input.  Else inv_norm is computed from the npart per-block maxima of each head
// (launch_normalize_partial_max) and written to inv_out
int launch_normalize_cast(const void* x, Strides3 xs, int dtype, void* y, const float* inv_norm, int B, int H, int N, int D,
                          hipStream_t stream, const unsigned int* partials, int npart, float* inv_out, int rep) {
    if (rep < 1) return FASTMAX_E_BAD_SHAPE;
    switch (dtype) {
        case FASTMAX_F32: return normalize_cast_t<float>(x, xs, y, inv_norm, B, H, N, D, stream, partials, npart, inv_out, rep);
        case FASTMAX_BF16: return normalize_cast_t<bf16_t>(x, xs, y, inv_norm, B, H, N, D, stream, partials, npart, inv_out, rep);
        case FASTMAX_F16: return normalize_cast_t<f16_t>(x, xs, y, inv_norm, B, H, N, D, stream, partials, npart, inv_out, rep);
    }
    return FASTMAX_E_BAD_DTYPE;
}

// Tokens per workgroup of the prologue's row passes.  rep > 1 (grouped-query form) shortens the block because every row is
// stored / read rep times.  ceil(256 / rep), not floor: ceil(N / tok) <= rep * ceil(N / 256) then holds for every rep, so the
// per-block records of the backward always fit a workspace sized for the expanded (B, G * rep, N) tensor.
int normalize_block_tokens(int rep) {
    if (rep <= 1) return 256;
    const int t = (256 + rep - 1) / rep;
    return t < 32 ? 32 : t;
}
static size_t normalize_backward_records(int B, int H, int nblk) {
    return (size_t)B * H * nblk * (sizeof(float) + sizeof(unsigned long long)) + 16;
}
size_t normalize_backward_workspace(int B, int H, int N) { return normalize_backward_records(B, H, (N + 255) / 256); }
// exact requirement of the grouped-query form (x has G heads, the gradient G * rep): one record per launched block
size_t normalize_backward_workspace_grouped(int B, int G, int rep, int N) {
    const int tok = normalize_block_tokens(rep);
    return normalize_backward_records(B, G, (N + tok - 1) / tok);
}

template <typename T>
static int normalize_bwd_t(const void* x, Strides3 xs, const void* gy, const float* inv_norm, void* gx, int B, int H, int N, int D,
                           void* ws, hipStream_t stream, int rep) {
    const int epl = (int)(16 / sizeof(T)), need = (D + epl - 1) / epl;
    if (need > 64) return FASTMAX_E_BAD_SHAPE;
    const int vec = nrm_vec_ok(x, xs, sizeof(T), D) && !((reinterpret_cast<uintptr_t>(gy) | reinterpret_cast<uintptr_t>(gx)) & 15);
    // rep > 1 (grouped-query form): every row reads rep gradient rows, so blocks take 256 / rep tokens (at least 32); the
    // partial records then number what the expanded tensor would need (the caller sizes the workspace for B, H rep, N)
    const int tok = normalize_block_tokens(rep);
    const int nblk = (N + tok - 1) / tok;
    // 8-byte records first (alignment), then the floats
    unsigned long long* part_best = reinterpret_cast<unsigned long long*>((reinterpret_cast<uintptr_t>(ws) + 7) & ~(uintptr_t)7);
    float* part_dot = reinterpret_cast<float*>(part_best + (size_t)B * H * nblk);
    const dim3 grid(nblk, B * H), block(256);
    static const bool two_pass_env = getenv("FASTMAX_NORMALIZE_BWD_TWO_PASS") != nullptr;  // the reduce + apply pair, for A/B runs
    const bool two_pass = two_pass_env && rep == 1;
#define CALL(L)                                                                                                                   \
    if (two_pass) {                                                                                                               \
        hipLaunchKernelGGL((normalize_bwd_reduce_kernel<T, L>), grid, block, 0, stream, x, xs, reinterpret_cast<const T*>(gy), H, N, D, \
                           part_dot, part_best, vec);                                                                             \
        hipLaunchKernelGGL((normalize_bwd_apply_kernel<T, L>), grid, block, 0, stream, x, xs, reinterpret_cast<const T*>(gy), inv_norm, \
                           part_dot, part_best, H, N, D, reinterpret_cast<T*>(gx), vec);                                          \
    } else {                                                                                                                      \
        hipLaunchKernelGGL((normalize_bwd_rows_kernel<T, L>), grid, block, 0, stream, x, xs, reinterpret_cast<const T*>(gy), inv_norm, H, N, \
                           D, reinterpret_cast<T*>(gx), part_dot, part_best, vec, rep, tok);                                      \
        hipLaunchKernelGGL((normalize_bwd_fix_kernel<T, L>), dim3(B * H), dim3(64), 0, stream, x, xs, reinterpret_cast<const T*>(gy),   \
                           inv_norm, part_dot, part_best, nblk, H, N, D, reinterpret_cast<T*>(gx), vec, rep);                     \
    }
    NRM_LPR_SWITCH(need, CALL)
#undef CALL
    return (int)hipGetLastError();
}
template <typename T>
static int normalize_fixadd_t(const void* x, Strides3 xs, const float* inv_norm, const float* part_dot, const int* nstar,
                              int nblk, int B, int H, int N, int D, void* gx, hipStream_t stream) {
    const int epl = (int)(16 / sizeof(T)), need = (D + epl - 1) / epl;
    if (need > 64) return FASTMAX_E_BAD_SHAPE;
    const int vec = nrm_vec_ok(x, xs, sizeof(T), D) && !(reinterpret_cast<uintptr_t>(gx) & 15);
#define CALL(L) hipLaunchKernelGGL((normalize_bwd_fixadd_kernel<T, L>), dim3(B * H), dim3(64), 0, stream, x, xs, inv_norm, part_dot, nstar, nblk, H, N, D, reinterpret_cast<T*>(gx), vec)
    NRM_LPR_SWITCH(need, CALL)
#undef CALL
    return (int)hipGetLastError();
}
// gx rows already hold inv (g - mean g): add the dL/dM term to the row that attains the max-norm (partials from the scan kernel)
int launch_normalize_fixadd(const void* x, Strides3 xs, int dtype, const float* inv_norm, const float* part_dot,
                            const int* nstar, int nblk, int B, int H, int N, int D, void* gx, hipStream_t stream) {
    switch (dtype) {
        case FASTMAX_F32: return normalize_fixadd_t<float>(x, xs, inv_norm, part_dot, nstar, nblk, B, H, N, D, gx, stream);
        case FASTMAX_BF16: return normalize_fixadd_t<bf16_t>(x, xs, inv_norm, part_dot, nstar, nblk, B, H, N, D, gx, stream);
        case FASTMAX_F16: return normalize_fixadd_t<f16_t>(x, xs, inv_norm, part_dot, nstar, nblk, B, H, N, D, gx, stream);
    }
    return FASTMAX_E_BAD_DTYPE;
}

int launch_normalize_backward(const void* x, Strides3 xs, int dtype, const void* gy, const float* inv_norm, void* gx, int B, int H,
                              int N, int D, void* ws, hipStream_t stream, int rep) {
    if (rep < 1) return FASTMAX_E_BAD_SHAPE;
    switch (dtype) {
        case FASTMAX_F32: return normalize_bwd_t<float>(x, xs, gy, inv_norm, gx, B, H, N, D, ws, stream, rep);
        case FASTMAX_BF16: return normalize_bwd_t<bf16_t>(x, xs, gy, inv_norm, gx, B, H, N, D, ws, stream, rep);
        case FASTMAX_F16: return normalize_bwd_t<f16_t>(x, xs, gy, inv_norm, gx, B, H, N, D, ws, stream, rep);
    }
    return FASTMAX_E_BAD_DTYPE;
}

}  // namespace fastmax
