// fastmax backward on the matrix cores (gfx950): dQ, dK, dV for p = 1, 2, masked and unmasked.
// Dense form of the reference's hand-derived gradients (attention_mechanisms/fastmax.py:383-691):
//   s_ij = a q_i.k_j,  P = f(s),  w_i = 1/g_i,  c_i = G_i.o_i,  u_ij = G_i.v_j
//   dS_ij = (u_ij - c_i) w_i f'(s_ij)        (only j <= i when causal)
//   dQ_i = a sum_j dS_ij k_j ;  dK_j = a sum_i dS_ij q_i ;  dV_j = sum_i P_ij w_i G_i
// Three launches: prep (c_i), dQ (one workgroup per 64 queries, loops over key tiles), dK/dV (one
// workgroup per 64 keys, loops over query tiles).  Scores are recomputed in both, so no atomics and the
// result is bitwise reproducible.  Operand handling as in fastmax_quad_mfma.hip.
#include "fastmax_mfma_common.h"

#include <type_traits>

namespace fastmax {

struct QuadBwdParams {
    const void *q, *k, *v, *o, *go;
    const float* g;
    Strides3 qs, ks, vs, gos;
    void *dq, *dk, *dv;
    float* c;                       // workspace (B,H,Nq)
    int H, Nq, Nk, D, causal, grad_dtype, o_dtype;
    float a;
    void* gt = nullptr;             // 32x32-tile kernels: prep also writes gt_i = w_i G_i (B,H,Nq,D, input dtype) and c_i w_i
};

// w_i G_i for the 32x32-tile kernels (fastmax_quad32_bwd.hip): with the row factor w_i = 1/g_i folded into the streamed
// operand, (u_ij - c_i) w_i = gt_i.v_j - c_i w_i leaves the score chain directly and dV_j = sum_i P_ij gt_i
template <typename TIN, int EPL>
__device__ __forceinline__ void store_scaled_piece(void* gt, int64_t elem, const float (&x)[EPL], float w) {
    if constexpr (sizeof(TIN) == 4) {
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(gt) + elem) = f32x4{x[0] * w, x[1] * w, x[2] * w, x[3] * w};
    } else if constexpr (InTraits<TIN>::NP == 1) {
        *reinterpret_cast<bf16x8*>(reinterpret_cast<uint16_t*>(gt) + elem) =
            cat4(to_bf16x4(f32x4{x[0] * w, x[1] * w, x[2] * w, x[3] * w}), to_bf16x4(f32x4{x[4] * w, x[5] * w, x[6] * w, x[7] * w}));
    } else {
        typedef _Float16 h8 __attribute__((ext_vector_type(8)));
        h8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (_Float16)(x[e] * w);
        *reinterpret_cast<h8*>(reinterpret_cast<_Float16*>(gt) + elem) = o;
    }
}

__device__ __forceinline__ float load_elem(const void* base, int dtype, int64_t idx) {
    if (dtype == FASTMAX_F32) return reinterpret_cast<const float*>(base)[idx];
    if (dtype == FASTMAX_BF16) return __uint_as_float(((uint32_t) reinterpret_cast<const uint16_t*>(base)[idx]) << 16);
    return (float)reinterpret_cast<const _Float16*>(base)[idx];
}

// c_i = G_i . o_i : one wave per query row.  grid = (ceil(Nq/4), B*H), block = 256
template <typename TIN>
__global__ __launch_bounds__(256) void bwd_prep_kernel(QuadBwdParams prm) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int bh = blockIdx.y, b = bh / prm.H, h = bh % prm.H;
    const int i = blockIdx.x * 4 + wave;
    if (i >= prm.Nq) return;
    const TIN* grow = row_ptr<TIN>(prm.go, prm.gos.sb, prm.gos.sh, prm.gos.sn, b, h, i);
    const int64_t ob = ((int64_t)bh * prm.Nq + i) * prm.D;
    float s = 0.f;
    for (int d = lane; d < prm.D; d += 64) s = fmaf(to_float(grow[d]), load_elem(prm.o, prm.o_dtype, ob + d), s);
    s = wave_sum(s);
    if (prm.gt) {
        const float w = 1.0f / prm.g[(int64_t)bh * prm.Nq + i];
        TIN* dst = reinterpret_cast<TIN*>(prm.gt) + ((int64_t)bh * prm.Nq + i) * prm.D;
        for (int d = lane; d < prm.D; d += 64) dst[d] = from_float<TIN>(to_float(grow[d]) * w);
        s *= w;
    }
    if (lane == 0) prm.c[(int64_t)bh * prm.Nq + i] = s;
}

// Vectorised c_i = G_i . o_i for head sizes with a power-of-two number of 16-byte pieces per row: LPR lanes per row,
// 256 / LPR rows per block, 16-byte loads of G and o.  grid = (ceil(Nq / (256/LPR)), B*H)
template <typename TIN, int LPR>
__global__ __launch_bounds__(256) void bwd_prep_vec_kernel(QuadBwdParams prm) {
    constexpr int EPL = InTraits<TIN>::EPL, RPB = 256 / LPR;
    const int tid = threadIdx.x, sub = tid % LPR;
    const int bh = blockIdx.y, b = bh / prm.H, h = bh % prm.H;
    const int i = blockIdx.x * RPB + tid / LPR;
    const bool live = i < prm.Nq;
    const int ic = live ? i : prm.Nq - 1;
    const TIN* grow = row_ptr<TIN>(prm.go, prm.gos.sb, prm.gos.sh, prm.gos.sn, b, h, ic);
    float x[EPL], y[EPL];
    piece_to_float<TIN>(*reinterpret_cast<const u32x4*>(grow + sub * EPL), x);
    const int64_t ob = ((int64_t)bh * prm.Nq + ic) * prm.D + sub * EPL;
    if (prm.o_dtype == FASTMAX_F32) {
#pragma unroll
        for (int e4 = 0; e4 < EPL / 4; ++e4) {
            const f32x4 o4 = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(prm.o) + ob + 4 * e4);
#pragma unroll
            for (int e = 0; e < 4; ++e) y[4 * e4 + e] = o4[e];
        }
    } else {      // 16-bit o has the input dtype (masked outputs; fastmax.py:100-106)
        if constexpr (sizeof(TIN) == 2) piece_to_float<TIN>(*reinterpret_cast<const u32x4*>(reinterpret_cast<const TIN*>(prm.o) + ob), y);
        else {
#pragma unroll
            for (int e = 0; e < EPL; ++e) y[e] = load_elem(prm.o, prm.o_dtype, ob + e);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) s = fmaf(x[e], y[e], s);
#pragma unroll
    for (int off = 1; off < LPR; off <<= 1) s += __shfl_xor(s, off, 64);
    if (prm.gt) {
        const float w = 1.0f / prm.g[(int64_t)bh * prm.Nq + ic];
        if (live) store_scaled_piece<TIN, EPL>(prm.gt, ((int64_t)bh * prm.Nq + i) * prm.D + sub * EPL, x, w);
        s *= w;
    }
    if (live && sub == 0) prm.c[(int64_t)bh * prm.Nq + i] = s;
}

template <typename TIN>
static void launch_prep(const QuadBwdParams& prm, int BH, hipStream_t stream) {
    constexpr int EPL = InTraits<TIN>::EPL;
    const int lpr = prm.D / EPL;
    const bool vec = prm.D % EPL == 0 && (lpr == 4 || lpr == 8 || lpr == 16 || lpr == 32) && (prm.gos.sn % EPL) == 0 &&
                     (prm.gos.sh % EPL) == 0 && (prm.gos.sb % EPL) == 0 && !(reinterpret_cast<uintptr_t>(prm.go) & 15) &&
                     !(reinterpret_cast<uintptr_t>(prm.o) & 15) && (sizeof(TIN) == 2 || prm.o_dtype == FASTMAX_F32);
    if (!vec) {
        hipLaunchKernelGGL((bwd_prep_kernel<TIN>), dim3((prm.Nq + 3) / 4, BH), dim3(256), 0, stream, prm);
        return;
    }
    const dim3 grid((prm.Nq + 256 / lpr - 1) / (256 / lpr), BH);
    switch (lpr) {
        case 4: hipLaunchKernelGGL((bwd_prep_vec_kernel<TIN, 4>), grid, dim3(256), 0, stream, prm); break;
        case 8: hipLaunchKernelGGL((bwd_prep_vec_kernel<TIN, 8>), grid, dim3(256), 0, stream, prm); break;
        case 16: hipLaunchKernelGGL((bwd_prep_vec_kernel<TIN, 16>), grid, dim3(256), 0, stream, prm); break;
        default: hipLaunchKernelGGL((bwd_prep_vec_kernel<TIN, 32>), grid, dim3(256), 0, stream, prm); break;
    }
}

// ---- dQ: grid = (ceil(Nq/64), B*H), block = 256, LDS = 4*NP*IMG ------------------------------------
template <int DP, int P, typename TIN>
__global__ __launch_bounds__(256, (DP == 64 || (DP == 128 && InTraits<TIN>::NP == 1)) ? 2 : 1) void bwd_dq_mfma_kernel(QuadBwdParams prm) {
    constexpr int NP = InTraits<TIN>::NP, EPL = InTraits<TIN>::EPL;
    constexpr int IMG = 64 * DP * 2;
    constexpr int QI = 0, GI = NP * IMG, KI = 2 * NP * IMG, VI = 3 * NP * IMG;
    constexpr int COLS = DP / EPL, RPP = 256 / COLS, NPASS = 64 / RPP;
    constexpr int KS = DP / 32, DT = DP / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q4 = lane >> 4;
    const int bh = blockIdx.y, b = bh / prm.H, h = bh % prm.H;
    const int D = prm.D, Nq = prm.Nq, Nk = prm.Nk;
    const bool causal = prm.causal != 0;
    const int qt = causal ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x;
    const int i0 = qt * 64;
    const TIN* qb = reinterpret_cast<const TIN*>(prm.q) + (int64_t)b * prm.qs.sb + (int64_t)h * prm.qs.sh;
    const TIN* kb = reinterpret_cast<const TIN*>(prm.k) + (int64_t)b * prm.ks.sb + (int64_t)h * prm.ks.sh;
    const TIN* vb = reinterpret_cast<const TIN*>(prm.v) + (int64_t)b * prm.vs.sb + (int64_t)h * prm.vs.sh;
    const TIN* gb = reinterpret_cast<const TIN*>(prm.go) + (int64_t)b * prm.gos.sb + (int64_t)h * prm.gos.sh;
    const int srow = tid / COLS, scol = tid % COLS;
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
        const int row = srow + ps * RPP;
        if constexpr (P == 2) stage_piece<DP, TIN>(smem, QI, row, scol, load_piece<TIN>(qb, prm.qs.sn, i0 + row, Nq, scol, D));
        stage_piece<DP, TIN>(smem, GI, row, scol, load_piece<TIN>(gb, prm.gos.sn, i0 + row, Nq, scol, D));
    }
    u32x4 rk[NPASS], rv[NPASS];
    const TileLoader<TIN, NPASS, RPP> kload(kb, prm.ks.sn, Nk, D, DP, srow, scol), vload(vb, prm.vs.sn, Nk, D, DP, srow, scol);
    auto issue = [&](int kt) {
        kload.load(kt, rk);
        vload.load(kt, rv);
    };
    const int nkt = causal ? qt + 1 : (Nk + 63) / 64;
    issue(0);
    __syncthreads();
    Frag<NP> qf[KS], gf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            if constexpr (P == 2) qf[ks].p[p] = ld_row8<DP>(smem, QI + p * IMG, 16 * w + r, 4 * ks + q4);
            gf[ks].p[p] = ld_row8<DP>(smem, GI + p * IMG, 16 * w + r, 4 * ks + q4);
        }
    const int qidx = i0 + 16 * w + r;
    const int qc = qidx < Nq ? qidx : Nq - 1;
    const float wi = 1.0f / prm.g[(int64_t)bh * Nq + qc];
    const float ci = prm.c[(int64_t)bh * Nq + qc];
    const float a = prm.a;
    f32x4 acc[DT];
#pragma unroll
    for (int mt = 0; mt < DT; ++mt) acc[mt] = f32x4{0, 0, 0, 0};

    constexpr int NPP = NP;                                  // parts of dS (one rounded part for bf16 problems)
    auto tile = [&](int kt, auto masked_tag) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        const bool diag = causal && kt == qt;
        Frag<NPP> df[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            f32x4 dt_[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int jt = 2 * s + e;
                f32x4 u = {0, 0, 0, 0}, sc = {0, 0, 0, 0};
                if (!(MASKED && diag && jt > w)) {
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        Frag<NP> vf, kf;
#pragma unroll
                        for (int p = 0; p < NP; ++p) {
                            vf.p[p] = ld_row8<DP>(smem, VI + p * IMG, 16 * jt + r, 4 * ks + q4);
                            if constexpr (P == 2) kf.p[p] = ld_row8<DP>(smem, KI + p * IMG, 16 * jt + r, 4 * ks + q4);
                        }
                        u = mfma_parts<NP, NP>(vf, gf[ks], u);                 // u[j][i] = v_j . G_i
                        if constexpr (P == 2) sc = mfma_parts<NP, NP>(kf, qf[ks], sc);
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float fp = (P == 2) ? 1.0f + a * sc[i] : 1.0f;
                    float dv_ = (u[i] - ci) * wi * fp;
                    if constexpr (MASKED) {
                        const int key = kt * 64 + 16 * jt + 4 * q4 + i;
                        const bool keep = key < Nk && (!causal || key <= qidx);
                        dv_ = keep ? dv_ : 0.f;
                    }
                    dt_[e][i] = dv_;
                }
            }
            if constexpr (NPP == 2) {
                bf16x4 h0, l0, h1, l1;
                split4(dt_[0], h0, l0);
                split4(dt_[1], h1, l1);
                df[s].p[0] = cat4(h0, h1);
                df[s].p[1] = cat4(l0, l1);
            } else {
                df[s].p[0] = cat4(to_bf16x4(dt_[0]), to_bf16x4(dt_[1]));
            }
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (!(MASKED && diag && 2 * s > w)) {
#pragma unroll
                for (int mt = 0; mt < DT; ++mt) {
                    Frag<NP> ktf;
#pragma unroll
                    for (int p = 0; p < NP; ++p) ktf.p[p] = ld_tr8<DP>(smem, KI + p * IMG, 32 * s, 16 * mt, lane);
                    acc[mt] = mfma_parts<NP, NPP>(ktf, df[s], acc[mt]);         // dQ^T[m][i] += K[j][m] dS[j][i]
                }
            }
        }
    };
    for (int kt = 0; kt < nkt; ++kt) {
        __syncthreads();
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            stage_piece<DP, TIN>(smem, KI, srow + ps * RPP, scol, rk[ps]);
            stage_piece<DP, TIN>(smem, VI, srow + ps * RPP, scol, rv[ps]);
        }
        if (kt + 1 < nkt) issue(kt + 1);
        __syncthreads();
        if ((causal && kt == qt) || (kt + 1) * 64 > Nk) tile(kt, std::true_type{});
        else tile(kt, std::false_type{});
    }
    __syncthreads();
    store_tile16<DP>(smem + KI + w * (16 * DP * 4), acc, a, lane, prm.dq, prm.grad_dtype,
                     ((int64_t)bh * Nq + i0 + 16 * w) * D, i0 + 16 * w, Nq, D);
}

// ---- dK, dV: grid = (ceil(Nk/64), B*H), block = 256, LDS = 4*NP*IMG + 512 --------------------------
template <int DP, int P, typename TIN>
__global__ __launch_bounds__(256, DP == 64 ? 2 : 1) void bwd_dkv_mfma_kernel(QuadBwdParams prm) {
    constexpr int NP = InTraits<TIN>::NP, EPL = InTraits<TIN>::EPL;
    constexpr int IMG = 64 * DP * 2;
    constexpr int QI = 0, GI = NP * IMG, KI = 2 * NP * IMG, VI = 3 * NP * IMG, WS = 4 * NP * IMG, CS = WS + 256;
    constexpr int COLS = DP / EPL, RPP = 256 / COLS, NPASS = 64 / RPP;
    constexpr int KS = DP / 32, DT = DP / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q4 = lane >> 4;
    const int bh = blockIdx.y, b = bh / prm.H, h = bh % prm.H;
    const int D = prm.D, Nq = prm.Nq, Nk = prm.Nk;
    const bool causal = prm.causal != 0;
    const int kt = blockIdx.x;                               // causal: low key tiles are the heavy ones and start first
    const int j0 = kt * 64;
    const TIN* qb = reinterpret_cast<const TIN*>(prm.q) + (int64_t)b * prm.qs.sb + (int64_t)h * prm.qs.sh;
    const TIN* kb = reinterpret_cast<const TIN*>(prm.k) + (int64_t)b * prm.ks.sb + (int64_t)h * prm.ks.sh;
    const TIN* vb = reinterpret_cast<const TIN*>(prm.v) + (int64_t)b * prm.vs.sb + (int64_t)h * prm.vs.sh;
    const TIN* gb = reinterpret_cast<const TIN*>(prm.go) + (int64_t)b * prm.gos.sb + (int64_t)h * prm.gos.sh;
    const int srow = tid / COLS, scol = tid % COLS;
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
        const int row = srow + ps * RPP;
        stage_piece<DP, TIN>(smem, KI, row, scol, load_piece<TIN>(kb, prm.ks.sn, j0 + row, Nk, scol, D));
        stage_piece<DP, TIN>(smem, VI, row, scol, load_piece<TIN>(vb, prm.vs.sn, j0 + row, Nk, scol, D));
    }
    u32x4 rq[NPASS], rg[NPASS];
    float rw = 0.f, rc = 0.f;
    const TileLoader<TIN, NPASS, RPP> qload(qb, prm.qs.sn, Nq, D, DP, srow, scol), gload(gb, prm.gos.sn, Nq, D, DP, srow, scol);
    auto issue = [&](int it) {
        qload.load(it, rq);
        gload.load(it, rg);
        if (tid < 64) {
            const int gi = it * 64 + tid, gc = gi < Nq ? gi : Nq - 1;
            rw = 1.0f / prm.g[(int64_t)bh * Nq + gc];
            rc = prm.c[(int64_t)bh * Nq + gc];
        }
    };
    const int it0 = causal ? kt : 0, nqt = (Nq + 63) / 64;
    issue(it0);
    const int kidx = j0 + 16 * w + r;
    const float a = prm.a;
    f32x4 dkacc[DT], dvacc[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t) { dkacc[t] = f32x4{0, 0, 0, 0}; dvacc[t] = f32x4{0, 0, 0, 0}; }

    constexpr int NPP = NP;
    auto tile = [&](int it, auto masked_tag) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        const bool diag = causal && it == kt;
        Frag<NPP> pwf[2], dsf[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            f32x4 pw[2], ds[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int itile = 2 * s + e;                 // 16 queries of the tile
                f32x4 sc = {0, 0, 0, 0}, u = {0, 0, 0, 0};
                if (!(MASKED && diag && itile < w)) {        // queries entirely before this wave's keys
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        Frag<NP> qf, gf, kf, vf;
#pragma unroll
                        for (int p = 0; p < NP; ++p) {
                            qf.p[p] = ld_row8<DP>(smem, QI + p * IMG, 16 * itile + r, 4 * ks + q4);
                            gf.p[p] = ld_row8<DP>(smem, GI + p * IMG, 16 * itile + r, 4 * ks + q4);
                            kf.p[p] = ld_row8<DP>(smem, KI + p * IMG, 16 * w + r, 4 * ks + q4);
                            vf.p[p] = ld_row8<DP>(smem, VI + p * IMG, 16 * w + r, 4 * ks + q4);
                        }
                        sc = mfma_parts<NP, NP>(qf, kf, sc);    // s[i][j]: rows = queries (regs), col = key (lane)
                        u = mfma_parts<NP, NP>(gf, vf, u);      // u[i][j] = G_i . v_j
                    }
                }
                const f32x4 w4 = *reinterpret_cast<const f32x4*>(smem + WS + (16 * itile + 4 * q4) * 4);
                const f32x4 c4 = *reinterpret_cast<const f32x4*>(smem + CS + (16 * itile + 4 * q4) * 4);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float sv = a * sc[i];
                    float pwv = poly_f<P>(sv) * w4[i];
                    float dsv = (u[i] - c4[i]) * w4[i] * poly_fprime<P>(sv);
                    if constexpr (MASKED) {
                        const int qi = it * 64 + 16 * itile + 4 * q4 + i;
                        const bool keep = qi < Nq && kidx < Nk && (!causal || qi >= kidx);
                        pwv = keep ? pwv : 0.f;
                        dsv = keep ? dsv : 0.f;
                    }
                    pw[e][i] = pwv;
                    ds[e][i] = dsv;
                }
            }
            if constexpr (NPP == 2) {
                bf16x4 h0, l0, h1, l1;
                split4(pw[0], h0, l0); split4(pw[1], h1, l1);
                pwf[s].p[0] = cat4(h0, h1); pwf[s].p[1] = cat4(l0, l1);
                split4(ds[0], h0, l0); split4(ds[1], h1, l1);
                dsf[s].p[0] = cat4(h0, h1); dsf[s].p[1] = cat4(l0, l1);
            } else {
                pwf[s].p[0] = cat4(to_bf16x4(pw[0]), to_bf16x4(pw[1]));
                dsf[s].p[0] = cat4(to_bf16x4(ds[0]), to_bf16x4(ds[1]));
            }
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (!(MASKED && diag && 2 * s + 1 < w)) {        // both 16-query halves of the k-step before the keys
#pragma unroll
                for (int t = 0; t < DT; ++t) {
                    Frag<NP> gtf, qtf;
#pragma unroll
                    for (int p = 0; p < NP; ++p) {
                        gtf.p[p] = ld_tr8<DP>(smem, GI + p * IMG, 32 * s, 16 * t, lane);
                        qtf.p[p] = ld_tr8<DP>(smem, QI + p * IMG, 32 * s, 16 * t, lane);
                    }
                    dvacc[t] = mfma_parts<NP, NPP>(gtf, pwf[s], dvacc[t]);      // dV^T[d][j] += G[i][d] P_ij w_i
                    dkacc[t] = mfma_parts<NP, NPP>(qtf, dsf[s], dkacc[t]);      // dK^T[m][j] += Q[i][m] dS_ij
                }
            }
        }
    };
    for (int it = it0; it < nqt; ++it) {
        __syncthreads();
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            stage_piece<DP, TIN>(smem, QI, srow + ps * RPP, scol, rq[ps]);
            stage_piece<DP, TIN>(smem, GI, srow + ps * RPP, scol, rg[ps]);
        }
        if (tid < 64) {
            reinterpret_cast<float*>(smem + WS)[tid] = rw;
            reinterpret_cast<float*>(smem + CS)[tid] = rc;
        }
        if (it + 1 < nqt) issue(it + 1);
        __syncthreads();
        if ((causal && it == kt) || (it + 1) * 64 > Nq || j0 + 64 > Nk) tile(it, std::true_type{});
        else tile(it, std::false_type{});
    }
    __syncthreads();
    store_tile16<DP>(smem + w * (16 * DP * 4), dkacc, a, lane, prm.dk, prm.grad_dtype,
                     ((int64_t)bh * Nk + j0 + 16 * w) * D, j0 + 16 * w, Nk, D);
    store_tile16<DP>(smem + DP * 256 + w * (16 * DP * 4), dvacc, 1.0f, lane, prm.dv, prm.grad_dtype,
                     ((int64_t)bh * Nk + j0 + 16 * w) * D, j0 + 16 * w, Nk, D);
}

template <int DP, int P, typename TIN>
static int launch_bwd_t(const QuadBwdParams& prm, int B, hipStream_t stream) {
    constexpr int NP = InTraits<TIN>::NP;
    constexpr int lds_q = 4 * NP * 64 * DP * 2, lds_kv = lds_q + 512;
    auto kq = bwd_dq_mfma_kernel<DP, P, TIN>;
    auto kkv = bwd_dkv_mfma_kernel<DP, P, TIN>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kq), hipFuncAttributeMaxDynamicSharedMemorySize, lds_q);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(kkv), hipFuncAttributeMaxDynamicSharedMemorySize, lds_kv);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const int BH = B * prm.H;
    launch_prep<TIN>(prm, BH, stream);
    hipLaunchKernelGGL(kq, dim3((prm.Nq + 63) / 64, BH), dim3(256), lds_q, stream, prm);
    hipLaunchKernelGGL(kkv, dim3((prm.Nk + 63) / 64, BH), dim3(256), lds_kv, stream, prm);
    return (int)hipGetLastError();
}
template <int P, typename TIN>
static int launch_bwd_d(const QuadBwdParams& prm, int B, hipStream_t stream) {
    if constexpr (InTraits<TIN>::NP == 1) {
        // head sizes 136 .. 256 (pythia-1b, Gemma, stablelm-3b): single-part operands only -- four 64 x 256 images are 128 KB
        if (prm.D > 128) return launch_bwd_t<256, P, TIN>(prm, B, stream);
    }
    return prm.D <= 64 ? launch_bwd_t<64, P, TIN>(prm, B, stream) : launch_bwd_t<128, P, TIN>(prm, B, stream);
}
template <typename TIN>
static int launch_bwd_p(const QuadBwdParams& prm, int B, int p, hipStream_t stream) {
    return p == 1 ? launch_bwd_d<1, TIN>(prm, B, stream) : launch_bwd_d<2, TIN>(prm, B, stream);
}

bool quad_mfma_bwd_supported(const fastmax_problem& p) {
    const int epl = p.in_dtype == FASTMAX_F32 ? 4 : 8;
    return (p.D % epl) == 0 && (p.D <= 128 || (p.D <= 256 && p.in_dtype == FASTMAX_BF16));
}

// 32x32-tile kernels (fastmax_quad32_bwd.hip) behind the same prep launch
int launch_bwd_quad32(const BwdArgs& a) {
    if (!quad32_bwd_supported(a.prob)) return FASTMAX_E_BAD_SHAPE;
    if (!quad32_bwd_layout_ok(a)) return launch_bwd_quad_mfma(a);
    if (a.workspace_bytes < bwd_quadratic_workspace(a.prob) || !a.workspace) return FASTMAX_E_WORKSPACE;
    QuadBwdParams prm{a.q, a.k, a.v, a.o, a.grad_o, a.g, a.qs, a.ks, a.vs, a.gos, a.dq, a.dk, a.dv,
                      reinterpret_cast<float*>(a.workspace), a.prob.H, a.prob.Nq, a.prob.Nk, a.prob.D, a.prob.causal,
                      a.prob.in_dtype, a.prob.out_dtype, a.prob.a};
    prm.gt = reinterpret_cast<char*>(a.workspace) + quad32_bwd_gt_offset(a.prob);
    switch (a.prob.in_dtype) {
        case FASTMAX_F32: launch_prep<float>(prm, a.prob.B * a.prob.H, a.stream); break;
        case FASTMAX_BF16: launch_prep<bf16_t>(prm, a.prob.B * a.prob.H, a.stream); break;
        case FASTMAX_F16: launch_prep<f16_t>(prm, a.prob.B * a.prob.H, a.stream); break;
        default: return FASTMAX_E_BAD_DTYPE;
    }
    const int e = (int)hipGetLastError();
    return e ? e : launch_bwd_quad32_main(a);
}

// c_i = G_i . o_i alone (the linear-time unmasked backward needs nothing else from the prep pass)
int launch_bwd_prep_c(const BwdArgs& a, float* cbuf) {
    QuadBwdParams prm{a.q, a.k, a.v, a.o, a.grad_o, a.g, a.qs, a.ks, a.vs, a.gos, a.dq, a.dk, a.dv,
                      cbuf, a.prob.H, a.prob.Nq, a.prob.Nk, a.prob.D, a.prob.causal, a.prob.in_dtype, a.prob.out_dtype, a.prob.a};
    switch (a.prob.in_dtype) {
        case FASTMAX_F32: launch_prep<float>(prm, a.prob.B * a.prob.H, a.stream); break;
        case FASTMAX_BF16: launch_prep<bf16_t>(prm, a.prob.B * a.prob.H, a.stream); break;
        case FASTMAX_F16: launch_prep<f16_t>(prm, a.prob.B * a.prob.H, a.stream); break;
        default: return FASTMAX_E_BAD_DTYPE;
    }
    return (int)hipGetLastError();
}

int launch_bwd_quad_mfma(const BwdArgs& a) {
    if (!quad_mfma_bwd_supported(a.prob)) return FASTMAX_E_BAD_SHAPE;
    if (a.workspace_bytes < sizeof(float) * (size_t)a.prob.B * a.prob.H * a.prob.Nq || !a.workspace) return FASTMAX_E_WORKSPACE;
    QuadBwdParams prm{a.q, a.k, a.v, a.o, a.grad_o, a.g, a.qs, a.ks, a.vs, a.gos, a.dq, a.dk, a.dv,
                      reinterpret_cast<float*>(a.workspace), a.prob.H, a.prob.Nq, a.prob.Nk, a.prob.D, a.prob.causal,
                      a.prob.in_dtype, a.prob.out_dtype, a.prob.a};
    switch (a.prob.in_dtype) {
        case FASTMAX_F32: return launch_bwd_p<float>(prm, a.prob.B, a.prob.p, a.stream);
        case FASTMAX_BF16: return launch_bwd_p<bf16_t>(prm, a.prob.B, a.prob.p, a.stream);
        case FASTMAX_F16: return launch_bwd_p<f16_t>(prm, a.prob.B, a.prob.p, a.stream);
    }
    return FASTMAX_E_BAD_DTYPE;
}

}  // namespace fastmax
