// Generic (vector-ALU) fastmax kernels for gfx950: every p / mask / dtype / head size <= 128.
//
//  * fwd_quadratic   o_i = sum_j f(s_ij) v_j / g_i evaluated tile by tile (one wave = 64
//                    queries, K/V tiles staged through LDS).  Same function as the
//                    reference's factorised sums (fastmax.py:184-322); O(N^2 D) work, used
//                    for p=2, for the unmasked / N_q != N_k cases and as the fallback.
//  * fwd_recurrent_p1  masked p=1 in linear time: the D x D running sum  S2 = sum k_j v_j^T
//                    is carried in registers (one value column per lane), the K running sum
//                    and the token count give the denominator (fastmax.py:236-241, 306-312
//                    without materialising the (N,D,D) outer products).
//  * bwd_quadratic   dQ, dK, dV from the dense form (== fastmax.py:383-691).
//  * normalize       linearmax prologue (fastmax.py:326-334 / fastmax_hack.py:38-43).
//
// These are the correctness-first paths; the matrix-core kernel for the headline shape is
// in fastmax_mfma.hip.
#include "fastmax_common.h"

namespace fastmax {

__device__ __forceinline__ void store_out(void* base, int dtype, int64_t idx, float val) {
    if (dtype == FASTMAX_F32) reinterpret_cast<float*>(base)[idx] = val;
    else if (dtype == FASTMAX_BF16) reinterpret_cast<uint16_t*>(base)[idx] = f32_to_bf16_bits(val);
    else reinterpret_cast<_Float16*>(base)[idx] = (_Float16)val;
}

struct QuadParams {
    const void *q, *k, *v;
    Strides3 qs, ks, vs;
    void* o;
    float* g;
    int B, H, Nq, Nk, D, causal, out_dtype;
    float a, g0;
};

// output columns a block of the vector-ALU tile kernels accumulates (per lane, in registers)
template <int DMAX> constexpr int acc_cols() { return DMAX > 128 ? 128 : DMAX; }

// ------------------------------------------------------------------------------------------
// forward, quadratic tiles.  grid = (ceil(Nq/64), B*H, DMAX / acc_cols), block = 64 (one wave, one query/lane)
// ------------------------------------------------------------------------------------------
template <typename T, int DMAX, int P>
__global__ __launch_bounds__(64) void fwd_quadratic_kernel(QuadParams prm) {
    constexpr int TQ = 64, TK = 32, QS = DMAX + 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* q_s = smem;                 // [TQ][QS]   own rows, pre-scaled by a = 1/nt
    float* k_s = q_s + TQ * QS;        // [TK][DMAX]
    float* v_s = k_s + TK * DMAX;      // [TK][DMAX]
    const int tid = threadIdx.x;
    const int bh = blockIdx.y, b = bh / prm.H, h = bh % prm.H;
    const int i0 = blockIdx.x * TQ, i = i0 + tid;
    const int D = prm.D;

    for (int idx = tid; idx < TQ * DMAX; idx += 64) {
        const int r = idx / DMAX, m = idx % DMAX, gi = i0 + r;
        float val = 0.f;
        if (gi < prm.Nq && m < D) val = prm.a * to_float(row_ptr<T>(prm.q, prm.qs.sb, prm.qs.sh, prm.qs.sn, b, h, gi)[m]);
        q_s[r * QS + m] = val;
    }
    // DMAX = 256: a block owns DA = 128 output columns (blockIdx.z picks the half; the scores use the whole row) -- a 256-entry
    // per-lane accumulator array does not survive register allocation intact on this toolchain (16-bit dK came out wrong in
    // one column), 128 entries is what every smaller head size uses
    constexpr int DA = acc_cols<DMAX>();
    const int d0 = blockIdx.z * DA;
    float acc[DA];
#pragma unroll
    for (int d = 0; d < DA; ++d) acc[d] = 0.f;
    float gsum = 0.f;
    const int jend = prm.causal ? min(prm.Nk, i0 + TQ) : prm.Nk;
    for (int j0 = 0; j0 < jend; j0 += TK) {
        __syncthreads();
        for (int idx = tid; idx < TK * DMAX; idx += 64) {
            const int r = idx / DMAX, m = idx % DMAX, gj = j0 + r;
            float kk = 0.f, vv = 0.f;
            if (gj < prm.Nk && m < D) {
                kk = to_float(row_ptr<T>(prm.k, prm.ks.sb, prm.ks.sh, prm.ks.sn, b, h, gj)[m]);
                vv = to_float(row_ptr<T>(prm.v, prm.vs.sb, prm.vs.sh, prm.vs.sn, b, h, gj)[m]);
            }
            k_s[idx] = kk;
            v_s[idx] = vv;
        }
        __syncthreads();
        const int jn = min(TK, jend - j0);
        for (int jj = 0; jj < jn; ++jj) {
            float s = 0.f;
#pragma unroll
            for (int m = 0; m < DMAX; m += 4) {
                const float4 qq = *reinterpret_cast<const float4*>(&q_s[tid * QS + m]);
                const float4 kk = *reinterpret_cast<const float4*>(&k_s[jj * DMAX + m]);
                s = fmaf(qq.x, kk.x, s); s = fmaf(qq.y, kk.y, s); s = fmaf(qq.z, kk.z, s); s = fmaf(qq.w, kk.w, s);
            }
            float pv = poly_f<P>(s);
            if (prm.causal && (j0 + jj) > i) pv = 0.f;
            gsum += pv;
#pragma unroll
            for (int d = 0; d < DA; d += 4) {
                const float4 vv = *reinterpret_cast<const float4*>(&v_s[jj * DMAX + d0 + d]);
                acc[d] = fmaf(pv, vv.x, acc[d]); acc[d + 1] = fmaf(pv, vv.y, acc[d + 1]);
                acc[d + 2] = fmaf(pv, vv.z, acc[d + 2]); acc[d + 3] = fmaf(pv, vv.w, acc[d + 3]);
            }
        }
    }
    // unmasked: rowsum(f) carries the constant N_k; the reference's constant is g0 (fastmax.py:271)
    const float gval = prm.causal ? gsum : gsum - (float)prm.Nk + prm.g0;
    const float inv = 1.0f / gval;
    __syncthreads();
#pragma unroll
    for (int d = 0; d < DA; ++d) q_s[tid * QS + d0 + d] = acc[d] * inv;
    if (i < prm.Nq && prm.g && d0 == 0) prm.g[(int64_t)bh * prm.Nq + i] = gval;
    __syncthreads();
    const int nrows = min(TQ, prm.Nq - i0);
    const int dn = min(DA, D - d0);                                // this block's live columns (<= 0: none)
    for (int idx = tid; idx < nrows * dn; idx += 64) {
        const int r = idx / dn, d = d0 + idx % dn;
        store_out(prm.o, prm.out_dtype, ((int64_t)bh * prm.Nq + i0 + r) * D + d, q_s[r * QS + d]);
    }
}

template <typename T, int DMAX>
static int launch_fwd_quadratic_t(const FwdArgs& a, const QuadParams& prm) {
    constexpr int TQ = 64, TK = 32, QS = DMAX + 4;
    const size_t lds = sizeof(float) * (TQ * QS + 2 * TK * DMAX);
    dim3 grid((a.prob.Nq + TQ - 1) / TQ, a.prob.B * a.prob.H, DMAX / acc_cols<DMAX>()), block(64);
    if constexpr (DMAX > 128) {
        static bool attr_set = false;
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fwd_quadratic_kernel<T, DMAX, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(fwd_quadratic_kernel<T, DMAX, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return (int)e;
            attr_set = true;
        }
    }
    if (a.prob.p == 1)
        hipLaunchKernelGGL((fwd_quadratic_kernel<T, DMAX, 1>), grid, block, lds, a.stream, prm);
    else
        hipLaunchKernelGGL((fwd_quadratic_kernel<T, DMAX, 2>), grid, block, lds, a.stream, prm);
    return (int)hipGetLastError();
}

template <typename T>
static int launch_fwd_quadratic_d(const FwdArgs& a, const QuadParams& prm) {
    const int D = a.prob.D;
    if (D <= 16) return launch_fwd_quadratic_t<T, 16>(a, prm);
    if (D <= 32) return launch_fwd_quadratic_t<T, 32>(a, prm);
    if (D <= 64) return launch_fwd_quadratic_t<T, 64>(a, prm);
    if (D <= 128) return launch_fwd_quadratic_t<T, 128>(a, prm);
    return launch_fwd_quadratic_t<T, 256>(a, prm);
}

int launch_fwd_quadratic(const FwdArgs& a) {
    QuadParams prm{a.q, a.k, a.v, a.qs, a.ks, a.vs, a.o, a.g, a.prob.B, a.prob.H, a.prob.Nq, a.prob.Nk,
                   a.prob.D, a.prob.causal, a.prob.out_dtype, a.prob.a, a.prob.g0};
    switch (a.prob.in_dtype) {
        case FASTMAX_F32: return launch_fwd_quadratic_d<float>(a, prm);
        case FASTMAX_BF16: return launch_fwd_quadratic_d<bf16_t>(a, prm);
        case FASTMAX_F16: return launch_fwd_quadratic_d<f16_t>(a, prm);
    }
    return FASTMAX_E_BAD_DTYPE;
}

// ------------------------------------------------------------------------------------------
// forward, masked p=1, linear time on the vector ALU.
// grid = B*H, block = DW*64 + 64 threads: DW = ceil(D/64) "column" waves (lane r of them
// carries column r of S2 = sum_j k_j v_j^T and S1[r] = sum_j v_j[r]) and one "g" wave that
// carries ksum = sum_j k_j and produces g_i = (i+1) + a q_i . ksum_i.
// ------------------------------------------------------------------------------------------
template <typename T, int DMAX>
__global__ __launch_bounds__((DMAX + 63) / 64 * 64 + 64) void fwd_recurrent_p1_kernel(QuadParams prm) {
    constexpr int TB = 32;                       // tokens staged per step
    constexpr int DW = (DMAX + 63) / 64;         // column waves
    constexpr int NT = DW * 64 + 64;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* q_s = smem;                 // [TB][DMAX]  (scaled by a)
    float* k_s = q_s + TB * DMAX;
    float* v_s = k_s + TB * DMAX;
    float* f_s = v_s + TB * DMAX;      // [TB][DMAX]  numerators
    float* g_s = f_s + TB * DMAX;      // [TB]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int bh = blockIdx.x, b = bh / prm.H, h = bh % prm.H;
    const int D = prm.D, N = prm.Nq;
    const bool gwave = (wave == DW);
    const int r = tid;                           // value column of a column-wave lane

    float S[DMAX];                               // column r of S2 (column waves only)
#pragma unroll
    for (int m = 0; m < DMAX; ++m) S[m] = 0.f;
    float s1 = 0.f;
    float ksum[DW];                              // g wave: lane holds m = lane + 64*u
#pragma unroll
    for (int u = 0; u < DW; ++u) ksum[u] = 0.f;

    for (int n0 = 0; n0 < N; n0 += TB) {
        __syncthreads();
        for (int idx = tid; idx < TB * DMAX; idx += NT) {
            const int t = idx / DMAX, m = idx % DMAX, gn = n0 + t;
            float qq = 0.f, kk = 0.f, vv = 0.f;
            if (gn < N && m < D) {
                qq = prm.a * to_float(row_ptr<T>(prm.q, prm.qs.sb, prm.qs.sh, prm.qs.sn, b, h, gn)[m]);
                kk = to_float(row_ptr<T>(prm.k, prm.ks.sb, prm.ks.sh, prm.ks.sn, b, h, gn)[m]);
                vv = to_float(row_ptr<T>(prm.v, prm.vs.sb, prm.vs.sh, prm.vs.sn, b, h, gn)[m]);
            }
            q_s[idx] = qq; k_s[idx] = kk; v_s[idx] = vv;
        }
        __syncthreads();
        const int tn = min(TB, N - n0);
        if (!gwave) {
            if (r < DMAX) {
                for (int t = 0; t < tn; ++t) {
                    const float vr = v_s[t * DMAX + r];
                    s1 += vr;
                    float out = 0.f;
#pragma unroll
                    for (int m = 0; m < DMAX; m += 4) {
                        const float4 kk = *reinterpret_cast<const float4*>(&k_s[t * DMAX + m]);
                        const float4 qq = *reinterpret_cast<const float4*>(&q_s[t * DMAX + m]);
                        S[m] = fmaf(kk.x, vr, S[m]); S[m + 1] = fmaf(kk.y, vr, S[m + 1]);
                        S[m + 2] = fmaf(kk.z, vr, S[m + 2]); S[m + 3] = fmaf(kk.w, vr, S[m + 3]);
                        out = fmaf(qq.x, S[m], out); out = fmaf(qq.y, S[m + 1], out);
                        out = fmaf(qq.z, S[m + 2], out); out = fmaf(qq.w, S[m + 3], out);
                    }
                    f_s[t * DMAX + r] = s1 + out;
                }
            }
        } else {
            for (int t = 0; t < tn; ++t) {
                float part = 0.f;
#pragma unroll
                for (int u = 0; u < DW; ++u) {
                    const int m = lane + 64 * u;
                    if (m < DMAX) {
                        ksum[u] += k_s[t * DMAX + m];
                        part = fmaf(q_s[t * DMAX + m], ksum[u], part);
                    }
                }
                part = wave_sum(part);
                if (lane == 0) g_s[t] = (float)(n0 + t + 1) + part;
            }
        }
        __syncthreads();
        for (int idx = tid; idx < tn * D; idx += NT) {
            const int t = idx / D, d = idx % D;
            store_out(prm.o, prm.out_dtype, ((int64_t)bh * N + n0 + t) * D + d, f_s[t * DMAX + d] / g_s[t]);
        }
        if (prm.g && tid < tn) prm.g[(int64_t)bh * N + n0 + tid] = g_s[tid];
    }
}

template <typename T, int DMAX>
static int launch_fwd_recurrent_t(const FwdArgs& a, const QuadParams& prm) {
    constexpr int TB = 32, NT = (DMAX + 63) / 64 * 64 + 64;
    const size_t lds = sizeof(float) * (4 * TB * DMAX + TB);
    hipLaunchKernelGGL((fwd_recurrent_p1_kernel<T, DMAX>), dim3(a.prob.B * a.prob.H), dim3(NT), lds, a.stream, prm);
    return (int)hipGetLastError();
}
template <typename T>
static int launch_fwd_recurrent_d(const FwdArgs& a, const QuadParams& prm) {
    const int D = a.prob.D;
    if (D <= 16) return launch_fwd_recurrent_t<T, 16>(a, prm);
    if (D <= 32) return launch_fwd_recurrent_t<T, 32>(a, prm);
    if (D <= 64) return launch_fwd_recurrent_t<T, 64>(a, prm);
    return launch_fwd_recurrent_t<T, 128>(a, prm);
}
int launch_fwd_recurrent_p1(const FwdArgs& a) {
    QuadParams prm{a.q, a.k, a.v, a.qs, a.ks, a.vs, a.o, a.g, a.prob.B, a.prob.H, a.prob.Nq, a.prob.Nk,
                   a.prob.D, a.prob.causal, a.prob.out_dtype, a.prob.a, a.prob.g0};
    switch (a.prob.in_dtype) {
        case FASTMAX_F32: return launch_fwd_recurrent_d<float>(a, prm);
        case FASTMAX_BF16: return launch_fwd_recurrent_d<bf16_t>(a, prm);
        case FASTMAX_F16: return launch_fwd_recurrent_d<f16_t>(a, prm);
    }
    return FASTMAX_E_BAD_DTYPE;
}

// ------------------------------------------------------------------------------------------
// backward, quadratic tiles (dense form of fastmax.py:383-691):
//   s_ij = a q_i.k_j,  P = f(s),  w_i = 1/g_i,  c_i = G_i.o_i,  u_ij = G_i.v_j
//   dS_ij = (u_ij - c_i) w_i f'(s_ij)           (masked: only j <= i)
//   dQ_i = a sum_j dS_ij k_j ;  dK_j = a sum_i dS_ij q_i ;  dV_j = sum_i P_ij w_i G_i
// ------------------------------------------------------------------------------------------
struct BwdParams {
    const void *q, *k, *v, *o, *go;
    const float* g;
    Strides3 qs, ks, vs, gos;
    void *dq, *dk, *dv;
    float* c;            // workspace (B,H,Nq): c_i = G_i . o_i
    int B, H, Nq, Nk, D, causal, out_dtype, o_dtype;
    float a;
};

__device__ __forceinline__ float load_any(const void* base, int dtype, int64_t idx) {
    if (dtype == FASTMAX_F32) return reinterpret_cast<const float*>(base)[idx];
    if (dtype == FASTMAX_BF16) return __uint_as_float(((uint32_t) reinterpret_cast<const uint16_t*>(base)[idx]) << 16);
    return (float)reinterpret_cast<const _Float16*>(base)[idx];
}

// rows of the streamed tile: 32, or 8 at DMAX = 256 (two own 64 x 260 float tiles are 130 KB of the 160 KB)
template <int DMAX> constexpr int bwd_stream_rows() { return DMAX > 128 ? 8 : 32; }

// dQ: one query per lane.  grid = (ceil(Nq/64), B*H), block = 64
template <typename T, int DMAX, int P>
__global__ __launch_bounds__(64) void bwd_dq_kernel(BwdParams prm) {
    constexpr int TQ = 64, TK = bwd_stream_rows<DMAX>(), QS = DMAX + 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* q_s = smem;                  // [TQ][QS] scaled by a
    float* G_s = q_s + TQ * QS;         // [TQ][QS]
    float* k_s = G_s + TQ * QS;         // [TK][DMAX]
    float* v_s = k_s + TK * DMAX;       // [TK][DMAX]
    const int tid = threadIdx.x;
    const int bh = blockIdx.y, b = bh / prm.H, h = bh % prm.H;
    const int i0 = blockIdx.x * TQ, i = i0 + tid, D = prm.D;
    for (int idx = tid; idx < TQ * DMAX; idx += 64) {
        const int r = idx / DMAX, m = idx % DMAX, gi = i0 + r;
        float qq = 0.f, gg = 0.f;
        if (gi < prm.Nq && m < D) {
            qq = prm.a * to_float(row_ptr<T>(prm.q, prm.qs.sb, prm.qs.sh, prm.qs.sn, b, h, gi)[m]);
            gg = to_float(row_ptr<T>(prm.go, prm.gos.sb, prm.gos.sh, prm.gos.sn, b, h, gi)[m]);
        }
        q_s[r * QS + m] = qq;
        G_s[r * QS + m] = gg;
    }
    __syncthreads();
    float w = 0.f, c = 0.f;
    if (i < prm.Nq) {
        w = 1.0f / prm.g[(int64_t)bh * prm.Nq + i];
        for (int d = 0; d < D; ++d)
            c = fmaf(G_s[tid * QS + d], load_any(prm.o, prm.o_dtype, ((int64_t)bh * prm.Nq + i) * D + d), c);
        prm.c[(int64_t)bh * prm.Nq + i] = c;
    }
    constexpr int DA = acc_cols<DMAX>();                           // see fwd_quadratic_kernel
    const int d0 = blockIdx.z * DA;
    float acc[DA];
#pragma unroll
    for (int d = 0; d < DA; ++d) acc[d] = 0.f;
    const int jend = prm.causal ? min(prm.Nk, i0 + TQ) : prm.Nk;
    for (int j0 = 0; j0 < jend; j0 += TK) {
        __syncthreads();
        for (int idx = tid; idx < TK * DMAX; idx += 64) {
            const int r = idx / DMAX, m = idx % DMAX, gj = j0 + r;
            float kk = 0.f, vv = 0.f;
            if (gj < prm.Nk && m < D) {
                kk = to_float(row_ptr<T>(prm.k, prm.ks.sb, prm.ks.sh, prm.ks.sn, b, h, gj)[m]);
                vv = to_float(row_ptr<T>(prm.v, prm.vs.sb, prm.vs.sh, prm.vs.sn, b, h, gj)[m]);
            }
            k_s[idx] = kk; v_s[idx] = vv;
        }
        __syncthreads();
        const int jn = min(TK, jend - j0);
        for (int jj = 0; jj < jn; ++jj) {
            float s = 0.f, u = 0.f;
#pragma unroll
            for (int m = 0; m < DMAX; m += 4) {
                const float4 qq = *reinterpret_cast<const float4*>(&q_s[tid * QS + m]);
                const float4 gg = *reinterpret_cast<const float4*>(&G_s[tid * QS + m]);
                const float4 kk = *reinterpret_cast<const float4*>(&k_s[jj * DMAX + m]);
                const float4 vv = *reinterpret_cast<const float4*>(&v_s[jj * DMAX + m]);
                s = fmaf(qq.x, kk.x, s); s = fmaf(qq.y, kk.y, s); s = fmaf(qq.z, kk.z, s); s = fmaf(qq.w, kk.w, s);
                u = fmaf(gg.x, vv.x, u); u = fmaf(gg.y, vv.y, u); u = fmaf(gg.z, vv.z, u); u = fmaf(gg.w, vv.w, u);
            }
            float dS = (u - c) * w * poly_fprime<P>(s);
            if (prm.causal && (j0 + jj) > i) dS = 0.f;
#pragma unroll
            for (int d = 0; d < DA; d += 4) {
                const float4 kk = *reinterpret_cast<const float4*>(&k_s[jj * DMAX + d0 + d]);
                acc[d] = fmaf(dS, kk.x, acc[d]); acc[d + 1] = fmaf(dS, kk.y, acc[d + 1]);
                acc[d + 2] = fmaf(dS, kk.z, acc[d + 2]); acc[d + 3] = fmaf(dS, kk.w, acc[d + 3]);
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int d = 0; d < DA; ++d) q_s[tid * QS + d0 + d] = acc[d] * prm.a;
    __syncthreads();
    const int nrows = min(TQ, prm.Nq - i0), dn = min(DA, D - d0);
    for (int idx = tid; idx < nrows * dn; idx += 64) {
        const int r = idx / dn, d = d0 + idx % dn;
        store_out(prm.dq, prm.out_dtype, ((int64_t)bh * prm.Nq + i0 + r) * D + d, q_s[r * QS + d]);
    }
}

// dK (MODE 0) / dV (MODE 1): one key per lane.  grid = (ceil(Nk/64), B*H), block = 64
template <typename T, int DMAX, int P, int MODE>
__global__ __launch_bounds__(64) void bwd_dkv_kernel(BwdParams prm) {
    constexpr int TJ = 64, TI = bwd_stream_rows<DMAX>(), QS = DMAX + 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* k_s = smem;                  // [TJ][QS] own keys, scaled by a
    float* v_s = k_s + TJ * QS;         // [TJ][QS] own values (dK only)
    float* q_s = v_s + TJ * QS;         // [TI][DMAX]
    float* G_s = q_s + TI * DMAX;       // [TI][DMAX]
    float* w_s = G_s + TI * DMAX;       // [TI]
    float* c_s = w_s + TI;              // [TI]
    const int tid = threadIdx.x;
    const int bh = blockIdx.y, b = bh / prm.H, h = bh % prm.H;
    const int j0 = blockIdx.x * TJ, j = j0 + tid, D = prm.D;
    for (int idx = tid; idx < TJ * DMAX; idx += 64) {
        const int r = idx / DMAX, m = idx % DMAX, gj = j0 + r;
        float kk = 0.f, vv = 0.f;
        if (gj < prm.Nk && m < D) {
            kk = prm.a * to_float(row_ptr<T>(prm.k, prm.ks.sb, prm.ks.sh, prm.ks.sn, b, h, gj)[m]);
            vv = to_float(row_ptr<T>(prm.v, prm.vs.sb, prm.vs.sh, prm.vs.sn, b, h, gj)[m]);
        }
        k_s[r * QS + m] = kk;
        v_s[r * QS + m] = vv;
    }
    constexpr int DA = acc_cols<DMAX>();                           // see fwd_quadratic_kernel
    const int d0 = blockIdx.z * DA;
    float acc[DA];
#pragma unroll
    for (int d = 0; d < DA; ++d) acc[d] = 0.f;
    const int ibeg = prm.causal ? j0 : 0;          // j0 is a multiple of 64, hence of TI
    for (int i0 = ibeg; i0 < prm.Nq; i0 += TI) {
        __syncthreads();
        for (int idx = tid; idx < TI * DMAX; idx += 64) {
            const int r = idx / DMAX, m = idx % DMAX, gi = i0 + r;
            float qq = 0.f, gg = 0.f;
            if (gi < prm.Nq && m < D) {
                qq = to_float(row_ptr<T>(prm.q, prm.qs.sb, prm.qs.sh, prm.qs.sn, b, h, gi)[m]);
                gg = to_float(row_ptr<T>(prm.go, prm.gos.sb, prm.gos.sh, prm.gos.sn, b, h, gi)[m]);
            }
            q_s[idx] = qq; G_s[idx] = gg;
        }
        if (tid < TI) {
            const int gi = i0 + tid;
            w_s[tid] = gi < prm.Nq ? 1.0f / prm.g[(int64_t)bh * prm.Nq + gi] : 0.f;
            c_s[tid] = gi < prm.Nq ? prm.c[(int64_t)bh * prm.Nq + gi] : 0.f;
        }
        __syncthreads();
        const int in = min(TI, prm.Nq - i0);
        for (int ii = 0; ii < in; ++ii) {
            float s = 0.f, u = 0.f;
#pragma unroll
            for (int m = 0; m < DMAX; m += 4) {
                const float4 kk = *reinterpret_cast<const float4*>(&k_s[tid * QS + m]);
                const float4 qq = *reinterpret_cast<const float4*>(&q_s[ii * DMAX + m]);
                s = fmaf(qq.x, kk.x, s); s = fmaf(qq.y, kk.y, s); s = fmaf(qq.z, kk.z, s); s = fmaf(qq.w, kk.w, s);
                if constexpr (MODE == 0) {
                    const float4 vv = *reinterpret_cast<const float4*>(&v_s[tid * QS + m]);
                    const float4 gg = *reinterpret_cast<const float4*>(&G_s[ii * DMAX + m]);
                    u = fmaf(gg.x, vv.x, u); u = fmaf(gg.y, vv.y, u); u = fmaf(gg.z, vv.z, u); u = fmaf(gg.w, vv.w, u);
                }
            }
            float coef;
            if constexpr (MODE == 0) coef = (u - c_s[ii]) * w_s[ii] * poly_fprime<P>(s);
            else coef = poly_f<P>(s) * w_s[ii];
            if (prm.causal && (i0 + ii) < j) coef = 0.f;
            const float* src = (MODE == 0) ? q_s : G_s;
#pragma unroll
            for (int d = 0; d < DA; d += 4) {
                const float4 xx = *reinterpret_cast<const float4*>(&src[ii * DMAX + d0 + d]);
                acc[d] = fmaf(coef, xx.x, acc[d]); acc[d + 1] = fmaf(coef, xx.y, acc[d + 1]);
                acc[d + 2] = fmaf(coef, xx.z, acc[d + 2]); acc[d + 3] = fmaf(coef, xx.w, acc[d + 3]);
            }
        }
    }
    __syncthreads();
    const float sc = (MODE == 0) ? prm.a : 1.0f;
#pragma unroll
    for (int d = 0; d < DA; ++d) k_s[tid * QS + d0 + d] = acc[d] * sc;
    __syncthreads();
    const int nrows = min(TJ, prm.Nk - j0), dn = min(DA, D - d0);
    void* dst = (MODE == 0) ? prm.dk : prm.dv;
    for (int idx = tid; idx < nrows * dn; idx += 64) {
        const int r = idx / dn, d = d0 + idx % dn;
        store_out(dst, prm.out_dtype, ((int64_t)bh * prm.Nk + j0 + r) * D + d, k_s[r * QS + d]);
    }
}

template <typename T, int DMAX, int P>
static int launch_bwd_tp(const BwdArgs& a, const BwdParams& prm) {
    constexpr int QS = DMAX + 4, TS = bwd_stream_rows<DMAX>();
    const size_t lds_q = sizeof(float) * (2 * 64 * QS + 2 * TS * DMAX);
    const size_t lds_kv = sizeof(float) * (2 * 64 * QS + 2 * TS * DMAX + 64);
    if constexpr (DMAX > 128) {        // above the 64 KB default
        static bool attr_set = false;
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(bwd_dq_kernel<T, DMAX, P>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_q);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(bwd_dkv_kernel<T, DMAX, P, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_kv);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(bwd_dkv_kernel<T, DMAX, P, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_kv);
            if (e != hipSuccess) return (int)e;
            attr_set = true;
        }
    }
    const int BH = a.prob.B * a.prob.H;
    constexpr int NZ = DMAX / acc_cols<DMAX>();
    hipLaunchKernelGGL((bwd_dq_kernel<T, DMAX, P>), dim3((a.prob.Nq + 63) / 64, BH, NZ), dim3(64), lds_q, a.stream, prm);
    hipLaunchKernelGGL((bwd_dkv_kernel<T, DMAX, P, 0>), dim3((a.prob.Nk + 63) / 64, BH, NZ), dim3(64), lds_kv, a.stream, prm);
    hipLaunchKernelGGL((bwd_dkv_kernel<T, DMAX, P, 1>), dim3((a.prob.Nk + 63) / 64, BH, NZ), dim3(64), lds_kv, a.stream, prm);
    return (int)hipGetLastError();
}
template <typename T, int DMAX>
static int launch_bwd_t(const BwdArgs& a, const BwdParams& prm) {
    return a.prob.p == 1 ? launch_bwd_tp<T, DMAX, 1>(a, prm) : launch_bwd_tp<T, DMAX, 2>(a, prm);
}
template <typename T>
static int launch_bwd_d(const BwdArgs& a, const BwdParams& prm) {
    const int D = a.prob.D;
    if (D <= 16) return launch_bwd_t<T, 16>(a, prm);
    if (D <= 32) return launch_bwd_t<T, 32>(a, prm);
    if (D <= 64) return launch_bwd_t<T, 64>(a, prm);
    if (D <= 128) return launch_bwd_t<T, 128>(a, prm);
    return launch_bwd_t<T, 256>(a, prm);
}
// c (B,H,Nq) floats; the 32x32-tile kernels add gt = w G (B,H,Nq,D) in the input dtype behind it (fastmax_quad_mfma_bwd.hip)
size_t quad32_bwd_gt_offset(const fastmax_problem& p) { return (sizeof(float) * (size_t)p.B * p.H * p.Nq + 255) & ~(size_t)255; }
size_t bwd_quadratic_workspace(const fastmax_problem& p) {
    if (!quad32_bwd_supported(p)) return sizeof(float) * (size_t)p.B * p.H * p.Nq;
    return quad32_bwd_gt_offset(p) + (size_t)p.B * p.H * p.Nq * p.D * (p.in_dtype == FASTMAX_F32 ? 4 : 2);
}

int launch_bwd_quadratic(const BwdArgs& a) {
    if (a.workspace_bytes < bwd_quadratic_workspace(a.prob) || !a.workspace) return FASTMAX_E_WORKSPACE;
    // o has the forward's out_dtype; grad_o and the three gradients have in_dtype (autograd casts
    // gradients to the dtype of the input they belong to)
    BwdParams prm{a.q, a.k, a.v, a.o, a.grad_o, a.g, a.qs, a.ks, a.vs, a.gos, a.dq, a.dk, a.dv,
                  reinterpret_cast<float*>(a.workspace), a.prob.B, a.prob.H, a.prob.Nq, a.prob.Nk, a.prob.D,
                  a.prob.causal, a.prob.in_dtype, a.prob.out_dtype, a.prob.a};
    switch (a.prob.in_dtype) {
        case FASTMAX_F32: return launch_bwd_d<float>(a, prm);
        case FASTMAX_BF16: return launch_bwd_d<bf16_t>(a, prm);
        case FASTMAX_F16: return launch_bwd_d<f16_t>(a, prm);
    }
    return FASTMAX_E_BAD_DTYPE;
}

// ------------------------------------------------------------------------------------------
// linearmax prologue (fastmax.py:326-334): one wave per token
// ------------------------------------------------------------------------------------------
// max over tokens of the squared centred norm.  One block walks TOK tokens of one head with 16-byte loads:
// a token row is spread over LPR lanes (LPR = 32 covers D <= 128 in fp32, D <= 256 in 16-bit; 64: D <= 256 in fp32), the running max
// stays in registers and each block issues ONE atomicMax (token-per-wave with an atomic each serialised on
// the head's word).  Rows must be 16-byte aligned; `vec` = 0 selects the scalar-load form for unaligned views.
template <typename T, int LPR>
__global__ __launch_bounds__(256) void normalize_max_kernel(const void* x, Strides3 xs, int H, int N, int D,
                                                            unsigned int* maxbits, int vec, int partial) {
    constexpr int TOK = 256, EPL = 16 / sizeof(T), RPB = 256 / LPR;
    __shared__ float wmax[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sub = tid % LPR, rgrp = tid / LPR;
    const int bh = blockIdx.y, b = bh / H, h = bh % H;
    const int n_begin = blockIdx.x * TOK, n_end = min(N, n_begin + TOK);
    float best = 0.f;
    for (int n = n_begin + rgrp; n < n_end; n += RPB) {
        const T* row = row_ptr<T>(x, xs.sb, xs.sh, xs.sn, b, h, n);
        float v[EPL];
        if (vec) {
            typedef unsigned int u4 __attribute__((ext_vector_type(4)));
            u4 raw = {0, 0, 0, 0};
            if (sub * EPL < D) raw = __builtin_nontemporal_load(reinterpret_cast<const u4*>(row + sub * EPL));
            const T* pv = reinterpret_cast<const T*>(&raw);
#pragma unroll
            for (int e = 0; e < EPL; ++e) v[e] = (sub * EPL + e) < D ? to_float(pv[e]) : 0.f;
        } else {
#pragma unroll
            for (int e = 0; e < EPL; ++e) v[e] = (sub * EPL + e) < D ? to_float(row[sub * EPL + e]) : 0.f;
        }
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) s += v[e];
#pragma unroll
        for (int off = 1; off < LPR; off <<= 1) s += __shfl_xor(s, off, 64);
        const float mean = s / (float)D;
        float nn = 0.f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const float c = (sub * EPL + e) < D ? v[e] - mean : 0.f;
            nn = fmaf(c, c, nn);
        }
#pragma unroll
        for (int off = 1; off < LPR; off <<= 1) nn += __shfl_xor(nn, off, 64);
        best = fmaxf(best, nn);
    }
    best = wave_max(best);
    if (lane == 0) wmax[wave] = best;
    __syncthreads();
    // squared norms are >= 0, so their float bit patterns order like unsigned integers
    if (tid == 0) {
        const unsigned int bits = __float_as_uint(fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3])));
        // partial: one word per block, combined by the consumer (no zeroing pass, no atomics); else one word per head
        if (partial) maxbits[(int64_t)bh * gridDim.x + blockIdx.x] = bits;
        else atomicMax(&maxbits[bh], bits);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void normalize_apply_kernel(const void* x, Strides3 xs, int H, int N, int D,
                                                              const unsigned int* maxbits, float* y,
                                                              float* inv_norm) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int bh = blockIdx.y, b = bh / H, h = bh % H;
    const int n = blockIdx.x * 4 + wave;
    const float inv = 1.0f / sqrtf(__uint_as_float(maxbits[bh]));
    if (blockIdx.x == 0 && threadIdx.x == 0 && inv_norm) inv_norm[bh] = inv;
    if (n >= N) return;
    const T* row = row_ptr<T>(x, xs.sb, xs.sh, xs.sn, b, h, n);
    float xv[4], s = 0.f;                                     // D <= 256: four columns per lane
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        xv[j] = lane + 64 * j < D ? to_float(row[lane + 64 * j]) : 0.f;
        s += xv[j];
    }
    const float mean = wave_sum(s) / (float)D;
    float* out = y + ((int64_t)bh * N + n) * D;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (lane + 64 * j < D) out[lane + 64 * j] = (xv[j] - mean) * inv;
}

static int rows_vec_ok(const void* x, Strides3 xs, size_t es, int D) {
    const int epl = (int)(16 / es);
    return (reinterpret_cast<uintptr_t>(x) % 16 == 0) && ((xs.sb * es) % 16 == 0) && ((xs.sh * es) % 16 == 0) &&
           ((xs.sn * es) % 16 == 0) && (D % epl == 0);
}

// lanes per token row: the smallest power of two that covers D with 16-byte pieces (4..32)
template <typename T>
static void launch_max_lpr(const void* x, Strides3 xs, int B, int H, int N, int D, unsigned int* maxbits, hipStream_t stream, int partial = 0) {
    const int epl = (int)(16 / sizeof(T)), need = (D + epl - 1) / epl;
    const int vec = rows_vec_ok(x, xs, sizeof(T), D);
    dim3 grid((N + 255) / 256, B * H), block(256);                 // TOK = 256 tokens per block
    if (need <= 4) hipLaunchKernelGGL((normalize_max_kernel<T, 4>), grid, block, 0, stream, x, xs, H, N, D, maxbits, vec, partial);
    else if (need <= 8) hipLaunchKernelGGL((normalize_max_kernel<T, 8>), grid, block, 0, stream, x, xs, H, N, D, maxbits, vec, partial);
    else if (need <= 16) hipLaunchKernelGGL((normalize_max_kernel<T, 16>), grid, block, 0, stream, x, xs, H, N, D, maxbits, vec, partial);
    else if (need <= 32) hipLaunchKernelGGL((normalize_max_kernel<T, 32>), grid, block, 0, stream, x, xs, H, N, D, maxbits, vec, partial);
    else hipLaunchKernelGGL((normalize_max_kernel<T, 64>), grid, block, 0, stream, x, xs, H, N, D, maxbits, vec, partial);
}

// zero the per-head max words with a kernel, not hipMemsetAsync: a memset node inside a captured HIP graph was observed
// to run unordered with the kernel that follows it (a quarter of the words zeroed after the atomics; ROCm 7.2)
__global__ void zero_words_kernel(unsigned int* p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0u;
}
static int zero_words(unsigned int* p, int n, hipStream_t stream) {
    hipLaunchKernelGGL(zero_words_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, p, n);
    return (int)hipGetLastError();
}

template <typename T>
static int launch_normalize_t(const void* x, Strides3 xs, float* y, float* inv_norm, int B, int H, int N, int D,
                              void* ws, hipStream_t stream) {
    unsigned int* maxbits = reinterpret_cast<unsigned int*>(ws);
    const int e = zero_words(maxbits, B * H, stream);
    if (e) return e;
    dim3 grid((N + 3) / 4, B * H), block(256);
    launch_max_lpr<T>(x, xs, B, H, N, D, maxbits, stream);
    hipLaunchKernelGGL((normalize_apply_kernel<T>), grid, block, 0, stream, x, xs, H, N, D, maxbits, y, inv_norm);
    return (int)hipGetLastError();
}
__global__ void normalize_finish_kernel(const unsigned int* maxbits, float* inv_norm, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) inv_norm[i] = 1.0f / sqrtf(__uint_as_float(maxbits[i]));
}
template <typename T>
static int launch_stats_t(const void* x, Strides3 xs, float* inv_norm, int B, int H, int N, int D, void* ws,
                          hipStream_t stream) {
    unsigned int* maxbits = reinterpret_cast<unsigned int*>(ws);
    const int e = zero_words(maxbits, B * H, stream);
    if (e) return e;
    launch_max_lpr<T>(x, xs, B, H, N, D, maxbits, stream);
    hipLaunchKernelGGL(normalize_finish_kernel, dim3((B * H + 255) / 256), dim3(256), 0, stream, maxbits, inv_norm, B * H);
    return (int)hipGetLastError();
}
// per-block maxima of the squared centred norm: partials[bh][ceil(N / 256)] (float bit patterns), no zeroing, no atomics
int launch_normalize_partial_max(const void* x, Strides3 xs, int dtype, unsigned int* partials, int B, int H, int N, int D,
                                 hipStream_t stream) {
    switch (dtype) {
        case FASTMAX_F32: launch_max_lpr<float>(x, xs, B, H, N, D, partials, stream, 1); break;
        case FASTMAX_BF16: launch_max_lpr<bf16_t>(x, xs, B, H, N, D, partials, stream, 1); break;
        case FASTMAX_F16: launch_max_lpr<f16_t>(x, xs, B, H, N, D, partials, stream, 1); break;
        default: return FASTMAX_E_BAD_DTYPE;
    }
    return (int)hipGetLastError();
}

int launch_normalize_stats(const void* x, Strides3 xs, int dtype, float* inv_norm, int B, int H, int N, int D,
                           void* workspace, hipStream_t stream) {
    switch (dtype) {
        case FASTMAX_F32: return launch_stats_t<float>(x, xs, inv_norm, B, H, N, D, workspace, stream);
        case FASTMAX_BF16: return launch_stats_t<bf16_t>(x, xs, inv_norm, B, H, N, D, workspace, stream);
        case FASTMAX_F16: return launch_stats_t<f16_t>(x, xs, inv_norm, B, H, N, D, workspace, stream);
    }
    return FASTMAX_E_BAD_DTYPE;
}

// The linearmax forward needs the statistic of TWO tensors (q and k): one launch over both (blockIdx.z picks the tensor),
// one word per 256-token block, and one small launch that folds the words into 1 / sqrt(max) -- 2 launches instead of 6
// (zero words, maxima with atomics, finish; twice), which at (1,32,16384,128) was 27 us of launch gaps beside 73 us of reads.
struct Max2Params {
    const void* x[2];
    Strides3 xs[2];
    unsigned long long* partials;      // [2][B*H][npart] keys: (bits of max ||row - mean||^2) << 32 | ~row
    int H, N, D, vec[2];
};
template <typename T, int LPR>
__global__ __launch_bounds__(256) void normalize_max2_kernel(Max2Params prm) {
    constexpr int TOK = 256, EPL = 16 / sizeof(T), RPB = 256 / LPR;
    __shared__ unsigned long long wmax[4];
    const int which = blockIdx.z;
    const void* x = prm.x[which];
    const Strides3 xs = prm.xs[which];
    const int vec = prm.vec[which];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sub = tid % LPR, rgrp = tid / LPR;
    const int bh = blockIdx.y, b = bh / prm.H, h = bh % prm.H;
    const int N = prm.N, D = prm.D;
    const int n_begin = blockIdx.x * TOK, n_end = min(N, n_begin + TOK);
    unsigned long long best = 0ull;                               // largest norm, lowest row on ties
    for (int n = n_begin + rgrp; n < n_end; n += RPB) {
        const T* row = row_ptr<T>(x, xs.sb, xs.sh, xs.sn, b, h, n);
        float v[EPL];
        if (vec) {
            typedef unsigned int u4 __attribute__((ext_vector_type(4)));
            u4 raw = {0, 0, 0, 0};
            if (sub * EPL < D) raw = *reinterpret_cast<const u4*>(row + sub * EPL);      // the scan re-reads it: default policy
            const T* pv = reinterpret_cast<const T*>(&raw);
#pragma unroll
            for (int e = 0; e < EPL; ++e) v[e] = (sub * EPL + e) < D ? to_float(pv[e]) : 0.f;
        } else {
#pragma unroll
            for (int e = 0; e < EPL; ++e) v[e] = (sub * EPL + e) < D ? to_float(row[sub * EPL + e]) : 0.f;
        }
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) s += v[e];
#pragma unroll
        for (int off = 1; off < LPR; off <<= 1) s += __shfl_xor(s, off, 64);
        const float mean = s / (float)D;
        float nn = 0.f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const float c = (sub * EPL + e) < D ? v[e] - mean : 0.f;
            nn = fmaf(c, c, nn);
        }
#pragma unroll
        for (int off = 1; off < LPR; off <<= 1) nn += __shfl_xor(nn, off, 64);
        const unsigned long long key = ((unsigned long long)__float_as_uint(nn) << 32) | (unsigned long long)(0xffffffffu - (unsigned)n);
        best = key > best ? key : best;
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long o = __shfl_xor(best, off, 64);
        best = o > best ? o : best;
    }
    if (lane == 0) wmax[wave] = best;
    __syncthreads();
    if (tid == 0) {
        unsigned long long m = wmax[0];
#pragma unroll
        for (int i = 1; i < 4; ++i) m = wmax[i] > m ? wmax[i] : m;
        prm.partials[((int64_t)which * gridDim.y + bh) * gridDim.x + blockIdx.x] = m;
    }
}
// inv[i] = 1 / sqrt(max over the npart keys of row i), nstar[i] = the row that attains it; rows = 2 B H (q heads, then k heads)
__global__ void normalize_finish_partials_kernel(const unsigned long long* partials, int npart, float* inv0, float* inv1, int nheads,
                                                 int* nstar0, int* nstar1) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * nheads) return;
    unsigned long long m = 0ull;
    for (int j = 0; j < npart; ++j) { const unsigned long long o = partials[(int64_t)i * npart + j]; m = o > m ? o : m; }
    (i < nheads ? inv0 : inv1)[i % nheads] = 1.0f / sqrtf(__uint_as_float((unsigned)(m >> 32)));
    int* ns = i < nheads ? nstar0 : nstar1;
    if (ns) ns[i % nheads] = (int)(0xffffffffu - (unsigned)(m & 0xffffffffull));
}
template <typename T>
static int launch_stats2_t(const void* x0, Strides3 s0, const void* x1, Strides3 s1, float* inv0, float* inv1, int B, int H, int N,
                           int D, void* ws, hipStream_t stream, int* nstar0, int* nstar1) {
    const int epl = (int)(16 / sizeof(T)), need = (D + epl - 1) / epl, npart = (N + 255) / 256;
    Max2Params prm{{x0, x1}, {s0, s1}, reinterpret_cast<unsigned long long*>(ws), H, N, D, {rows_vec_ok(x0, s0, sizeof(T), D), rows_vec_ok(x1, s1, sizeof(T), D)}};
    dim3 grid(npart, B * H, 2), block(256);
    if (need <= 4) hipLaunchKernelGGL((normalize_max2_kernel<T, 4>), grid, block, 0, stream, prm);
    else if (need <= 8) hipLaunchKernelGGL((normalize_max2_kernel<T, 8>), grid, block, 0, stream, prm);
    else if (need <= 16) hipLaunchKernelGGL((normalize_max2_kernel<T, 16>), grid, block, 0, stream, prm);
    else if (need <= 32) hipLaunchKernelGGL((normalize_max2_kernel<T, 32>), grid, block, 0, stream, prm);
    else hipLaunchKernelGGL((normalize_max2_kernel<T, 64>), grid, block, 0, stream, prm);
    hipLaunchKernelGGL(normalize_finish_partials_kernel, dim3((2 * B * H + 255) / 256), dim3(256), 0, stream,
                       reinterpret_cast<const unsigned long long*>(ws), npart, inv0, inv1, B * H, nstar0, nstar1);
    return (int)hipGetLastError();
}
int launch_normalize_stats2(const void* x0, Strides3 s0, const void* x1, Strides3 s1, int dtype, float* inv0, float* inv1, int B, int H,
                            int N, int D, void* ws, hipStream_t stream, int* nstar0, int* nstar1) {
    switch (dtype) {
        case FASTMAX_F32: return launch_stats2_t<float>(x0, s0, x1, s1, inv0, inv1, B, H, N, D, ws, stream, nstar0, nstar1);
        case FASTMAX_BF16: return launch_stats2_t<bf16_t>(x0, s0, x1, s1, inv0, inv1, B, H, N, D, ws, stream, nstar0, nstar1);
        case FASTMAX_F16: return launch_stats2_t<f16_t>(x0, s0, x1, s1, inv0, inv1, B, H, N, D, ws, stream, nstar0, nstar1);
    }
    return FASTMAX_E_BAD_DTYPE;
}

int launch_normalize(const void* x, Strides3 xs, int dtype, float* y, float* inv_norm, int B, int H, int N, int D,
                     void* workspace, hipStream_t stream) {
    switch (dtype) {
        case FASTMAX_F32: return launch_normalize_t<float>(x, xs, y, inv_norm, B, H, N, D, workspace, stream);
        case FASTMAX_BF16: return launch_normalize_t<bf16_t>(x, xs, y, inv_norm, B, H, N, D, workspace, stream);
        case FASTMAX_F16: return launch_normalize_t<f16_t>(x, xs, y, inv_norm, B, H, N, D, workspace, stream);
    }
    return FASTMAX_E_BAD_DTYPE;
}

}  // namespace fastmax
