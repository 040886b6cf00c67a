// fastmax p=1 masked forward for bf16 tensors, linear in N -- every sum on the matrix cores.
//
// Same chunked scan as fastmax_mfma.hip / fastmax_mfma_gen.hip, specialised for exact single-part bf16 operands:
// the "ones" terms that the fp32 kernels keep on the vector ALU ride on spare MFMA tiles here
//     S1   = sum_j v_j      = (1^T V)      one extra A fragment of ones against the V fragments of step (4)
//     ksum = sum_j k_j      = (K^T 1)      the K^T fragments of step (4) against a ones B fragment
//     a q_i.ksum_prev                       one extra 16-row tile (row D = a*ksum) of the (a S2)^T image in step (3)
// (products with 1.0 are exact, accumulation is fp32), so staging is a plain 16-byte copy global -> LDS, no per-chunk
// partial-sum arrays exist and the workgroup needs 45 KB of LDS at D <= 64: three workgroups per CU.
// NORM fuses the linearmax prologue (fastmax_hack.py:38-43) into staging as in the generic kernel.
#include "fastmax_mfma_common.h"

#include <cstdlib>

namespace fastmax {

struct Bf16Params {
    const void *q, *k, *v;
    Strides3 qs, ks, vs;
    void* o;
    float* g;
    const float *qscale, *kscale;
    int H, N, D;
    float a;
    const float* state;
    int nseg, cps;
};

// LEAN >= 1: the (a S2)^T state image is ONE rounded bf16 part instead of hi + lo: its rounding (2^-9 relative on entries
// whose products with q are summed over D terms, in a term that is itself ~1/8 of the numerator) is far below the bf16
// rounding of the result.  LEAN == 1 also carries P = 1 + a s as one part (the choice the tile kernels make for
// bf16 -> bf16 problems); that one shows on the first rows of a sequence, where the in-chunk sum is the whole numerator.
template <int DP, bool NORM, int LEAN>
__global__ __launch_bounds__(256, DP == 64 ? 2 : 1) void fwd_p1_mfma_bf16_kernel(Bf16Params prm) {
    using TIN = bf16_t;
    constexpr int EPL = 8, C = 64, IMG = C * DP * 2, SIMG = (DP + 16) * DP * 2;
    constexpr int SP = LEAN ? 1 : 2, PP = LEAN == 1 ? 1 : 2;      // parts of the state image / of P
    constexpr int QI = 0, KI = IMG, VI = 2 * IMG, S2I = 3 * IMG, S1V = S2I + SP * SIMG;
    constexpr int COLS = DP / EPL, RPP = 256 / COLS, NPASS = C / RPP;
    constexpr int KS = DP / 32, DT = DP / 16, MT = DP / 16, NSL = DP / 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q4 = lane >> 4;
    const int bh = blockIdx.x / prm.nseg, seg = blockIdx.x - bh * prm.nseg;
    const int b = bh / prm.H, h = bh % prm.H;
    const int N = prm.N, D = prm.D;
    const float a = prm.a;
    const TIN* qb = reinterpret_cast<const TIN*>(prm.q) + (int64_t)b * prm.qs.sb + (int64_t)h * prm.qs.sh;
    const TIN* kb = reinterpret_cast<const TIN*>(prm.k) + (int64_t)b * prm.ks.sb + (int64_t)h * prm.ks.sh;
    const TIN* vb = reinterpret_cast<const TIN*>(prm.v) + (int64_t)b * prm.vs.sb + (int64_t)h * prm.vs.sh;
    float qsc = 1.f, ksc = 1.f;
    if constexpr (NORM) { qsc = prm.qscale[bh]; ksc = prm.kscale[bh]; }
    const float invD = 1.0f / (float)D;
    const int srow = tid / COLS, scol = tid % COLS;
    const bool colok = scol * EPL < D;
    const float qsc_c = colok ? qsc : 0.f, ksc_c = colok ? ksc : 0.f;

    u32x4 rq[NPASS], rk[NPASS], rv[NPASS];
    // buffer-descriptor loads (non-temporal): rows past N and padded head columns arrive as zeros, no address arithmetic
    const BufTileLoader<TIN, NPASS, RPP, 2> qload(qb, prm.qs.sn, N, D, srow, scol), kload(kb, prm.ks.sn, N, D, srow, scol),
        vload(vb, prm.vs.sn, N, D, srow, scol);
    auto issue = [&](int c) {
        qload.load(c, rq);
        kload.load(c, rk);
        vload.load(c, rv);
    };
    // A / B fragment of ones in row / column 0 (lanes r == 0), zeros elsewhere
    bf16x8 ones;
#pragma unroll
    for (int i = 0; i < 8; ++i) ones[i] = (__bf16)(r == 0 ? 1.0f : 0.0f);

    const int nchunks = (N + C - 1) / C;
    const int c_begin = seg * prm.cps, c_end = min(nchunks, c_begin + prm.cps);
    f32x4 s2acc[NSL][MT];          // S2[16mt + 4q4 + reg][16(w + 4sl) + r]
    f32x4 s1acc[NSL];              // row 0 (q4 == 0, reg 0): S1[16(w + 4sl) + r]
    f32x4 ksacc[NSL];              // column 0 (r == 0): ksum[16(w + 4sl) + 4q4 + reg]
    auto publish = [&]() {
#pragma unroll
        for (int sl = 0; sl < NSL; ++sl) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int off = img_off<DP>(16 * (w + 4 * sl) + r, 2 * mt + (q4 >> 1)) + ((q4 & 1) << 3);
                if constexpr (LEAN) {
                    *reinterpret_cast<bf16x4*>(smem + S2I + off) = to_bf16x4(s2acc[sl][mt] * a);
                } else {
                    bf16x4 hi, lo;
                    split4(s2acc[sl][mt] * a, hi, lo);
                    *reinterpret_cast<bf16x4*>(smem + S2I + off) = hi;
                    *reinterpret_cast<bf16x4*>(smem + S2I + SIMG + off) = lo;
                }
            }
            if (r == 0) {                                        // image row DP = a * ksum, columns m of tile w + 4sl
                const int off = img_off<DP>(DP, 2 * (w + 4 * sl) + (q4 >> 1)) + ((q4 & 1) << 3);
                if constexpr (LEAN) {
                    *reinterpret_cast<bf16x4*>(smem + S2I + off) = to_bf16x4(ksacc[sl] * a);
                } else {
                    bf16x4 hi, lo;
                    split4(ksacc[sl] * a, hi, lo);
                    *reinterpret_cast<bf16x4*>(smem + S2I + off) = hi;
                    *reinterpret_cast<bf16x4*>(smem + S2I + SIMG + off) = lo;
                }
            }
            if (q4 == 0) reinterpret_cast<float*>(smem + S1V)[16 * (w + 4 * sl) + r] = s1acc[sl][0];
        }
    };
    for (int i = tid; i < (SP * SIMG) / 16; i += 256) *reinterpret_cast<f32x4*>(smem + S2I + 16 * i) = f32x4{0, 0, 0, 0};
    if (tid < DP) reinterpret_cast<float*>(smem + S1V)[tid] = 0.f;
#pragma unroll
    for (int sl = 0; sl < NSL; ++sl) {
        s1acc[sl] = f32x4{0, 0, 0, 0};
        ksacc[sl] = f32x4{0, 0, 0, 0};
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) s2acc[sl][mt] = f32x4{0, 0, 0, 0};
    }
    if (seg > 0) {
        __syncthreads();                                         // zero fill done before the partial overwrite
        const float* rec = prm.state + ((int64_t)bh * (prm.nseg - 1) + (seg - 1)) * (DP * DP + 2 * DP);
#pragma unroll
        for (int sl = 0; sl < NSL; ++sl) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int i = 0; i < 4; ++i) s2acc[sl][mt][i] = rec[(16 * mt + 4 * q4 + i) * DP + 16 * (w + 4 * sl) + r];
            if (q4 == 0) s1acc[sl][0] = rec[DP * DP + 16 * (w + 4 * sl) + r];
            if (r == 0)
#pragma unroll
                for (int i = 0; i < 4; ++i) ksacc[sl][i] = rec[DP * DP + DP + 16 * (w + 4 * sl) + 4 * q4 + i];
        }
        publish();
    }
    issue(c_begin);
    __syncthreads();

    for (int c = c_begin; c < c_end; ++c) {
        const int n0 = c * C;
        // ---- (a) staging: plain copies (NORM: mean-centred and scaled Q, K rows) ----------------------------
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int row = srow + ps * RPP;
            if constexpr (NORM) {
                float xq[EPL], xk[EPL];
                piece_to_float<TIN>(rq[ps], xq);
                piece_to_float<TIN>(rk[ps], xk);
                float sq = 0.f, sk = 0.f;
#pragma unroll
                for (int e = 0; e < EPL; ++e) { sq += xq[e]; sk += xk[e]; }
                // (x - mean) * scale as one fma per element; a row past N was loaded as zeros (mean 0 -> 0), a padded head
                // column has its scale zeroed (qsc_c / ksc_c)
                const float nmq = -rowgroup_allsum_dpp<COLS>(sq) * invD * qsc_c, nmk = -rowgroup_allsum_dpp<COLS>(sk) * invD * ksc_c;
#pragma unroll
                for (int e = 0; e < EPL; ++e) {
                    xq[e] = fmaf(xq[e], qsc_c, nmq);
                    xk[e] = fmaf(xk[e], ksc_c, nmk);
                }
                stage_floats<DP, EPL, 1>(smem, QI, row, scol, xq);
                stage_floats<DP, EPL, 1>(smem, KI, row, scol, xk);
            } else {
                *reinterpret_cast<u32x4*>(smem + QI + img_off<DP>(row, scol)) = rq[ps];
                *reinterpret_cast<u32x4*>(smem + KI + img_off<DP>(row, scol)) = rk[ps];
            }
            *reinterpret_cast<u32x4*>(smem + VI + img_off<DP>(row, scol)) = rv[ps];
        }
        if (c + 1 < c_end) issue(c + 1);
        __syncthreads();                                             // B1
        // ---- phase A -----------------------------------------------------------------------------------------
        const int qi = 16 * w + r;
        bf16x8 qf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = ld_row8<DP>(smem, QI, qi, 4 * ks + q4);
        f32x4 oacc[DT];
        f32x4 qkacc = {0, 0, 0, 0};
#pragma unroll
        for (int dt = 0; dt <= DT; ++dt) {                           // tile DT is the (a ksum) row
            f32x4 acc = {0, 0, 0, 0};
            if (dt < DT) acc = *reinterpret_cast<const f32x4*>(smem + S1V + (16 * dt + 4 * q4) * 4);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                acc = mfma(*reinterpret_cast<const bf16x8*>(smem + S2I + img_off<DP>(16 * dt + r, 4 * ks + q4)), qf[ks], acc);
                if constexpr (!LEAN)
                    acc = mfma(*reinterpret_cast<const bf16x8*>(smem + S2I + SIMG + img_off<DP>(16 * dt + r, 4 * ks + q4)), qf[ks], acc);
            }
            if (dt < DT) oacc[dt < DT ? dt : 0] = acc;
            else qkacc = acc;
        }
        const float qk = __shfl(qkacc[0], r, 64);                    // row 0 of the extra tile lives in lanes q4 == 0
        float gsum = 0.f;
        Frag<PP> pf[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            f32x4 pt[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int jt = 2 * s + e;
                f32x4 sc = {0, 0, 0, 0};
                if (jt <= w) {
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) sc = mfma(ld_row8<DP>(smem, KI, 16 * jt + r, 4 * ks + q4), qf[ks], sc);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bool keep = (jt < w) || (jt == w && (4 * q4 + i) <= r);
                    const float sv = keep ? a * sc[i] : 0.f;
                    gsum += sv;
                    pt[e][i] = keep ? 1.0f + sv : 0.f;
                }
            }
            if constexpr (PP == 1) {
                pf[s].p[0] = cat4(to_bf16x4(pt[0]), to_bf16x4(pt[1]));
            } else {
                bf16x4 h0, l0, h1, l1;
                split4(pt[0], h0, l0);
                split4(pt[1], h1, l1);
                pf[s].p[0] = cat4(h0, h1);
                pf[s].p[1] = cat4(l0, l1);
            }
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (2 * s <= w) {
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const bf16x8 vf = ld_tr8<DP>(smem, VI, 32 * s, 16 * dt, lane);
                    oacc[dt] = mfma(vf, pf[s].p[0], oacc[dt]);
                    if constexpr (PP == 2) oacc[dt] = mfma(vf, pf[s].p[1], oacc[dt]);
                }
            }
        }
        gsum += __shfl_xor(gsum, 16, 64);
        gsum += __shfl_xor(gsum, 32, 64);
        const int gi = n0 + qi;
        const float gval = (float)(gi + 1) + qk + gsum;
        if (gi < N && prm.g && q4 == 0) prm.g[(int64_t)bh * N + gi] = gval;
        store_tile16_private<DP, 2>(smem + QI + 16 * w * (2 * DP), nullptr, oacc, 1.0f / gval, lane, prm.o, FASTMAX_BF16,
                                    ((int64_t)bh * N + n0 + 16 * w) * D, n0 + 16 * w, N, D);
        // ---- phase B: S2 += K^T V, S1 += 1^T V, ksum += K^T 1 ------------------------------------------------
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int sl = 0; sl < NSL; ++sl) {
                const bf16x8 vf = ld_tr8<DP>(smem, VI, 32 * s, 16 * (w + 4 * sl), lane);
                s1acc[sl] = mfma(ones, vf, s1acc[sl]);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const bf16x8 kf = ld_tr8<DP>(smem, KI, 32 * s, 16 * mt, lane);
                    s2acc[sl][mt] = mfma(kf, vf, s2acc[sl][mt]);
                    if (mt == w + 4 * sl) ksacc[sl] = mfma(kf, ones, ksacc[sl]);      // wave-uniform
                }
            }
        }
        __syncthreads();                                             // B2
        if (c + 1 < c_end) publish();
    }
}

// ------------------------------------------------------------------------------------------------------------------
// D in (64, 128]: eight waves per workgroup (one workgroup per CU).  Wave (qt, dh): query tile qt = w & 3 of the chunk,
// output columns [64 dh, 64 dh + 64).  Scores are computed by both waves of a query tile (cheap next to the D^2 terms);
// the 128 x 128 state is spread over the eight waves (value-column slab 16 w each), so no wave holds more than 32
// accumulator registers of it and the kernel runs two waves per SIMD instead of one.
// ------------------------------------------------------------------------------------------------------------------
template <bool NORM>
__global__ __launch_bounds__(512, 2) void fwd_p1_mfma_bf16_d128_kernel(Bf16Params prm) {
    using TIN = bf16_t;
    constexpr int DP = 128, EPL = 8, C = 64, IMG = C * DP * 2, SIMG = (DP + 16) * DP * 2;
    constexpr int QI = 0, KI = IMG, VI = 2 * IMG, S2I = 3 * IMG, S1V = S2I + SIMG, OST = S1V + DP * 4;     // state image: one bf16 part
    constexpr int COLS = DP / EPL, RPP = 512 / COLS, NPASS = C / RPP;      // 16 lanes per row, 32 rows per pass, 2 passes
    constexpr int KS = DP / 32, MT = DP / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qt = w & 3, dh = w >> 2;
    const int r = lane & 15, q4 = lane >> 4;
    const int bh = blockIdx.x / prm.nseg, seg = blockIdx.x - bh * prm.nseg;
    const int b = bh / prm.H, h = bh % prm.H;
    const int N = prm.N, D = prm.D;
    const float a = prm.a;
    const TIN* qb = reinterpret_cast<const TIN*>(prm.q) + (int64_t)b * prm.qs.sb + (int64_t)h * prm.qs.sh;
    const TIN* kb = reinterpret_cast<const TIN*>(prm.k) + (int64_t)b * prm.ks.sb + (int64_t)h * prm.ks.sh;
    const TIN* vb = reinterpret_cast<const TIN*>(prm.v) + (int64_t)b * prm.vs.sb + (int64_t)h * prm.vs.sh;
    float qsc = 1.f, ksc = 1.f;
    if constexpr (NORM) { qsc = prm.qscale[bh]; ksc = prm.kscale[bh]; }
    const float invD = 1.0f / (float)D;
    const int srow = tid / COLS, scol = tid % COLS;
    const bool colok = scol * EPL < D;
    const float qsc_c = colok ? qsc : 0.f, ksc_c = colok ? ksc : 0.f;

    u32x4 rq[NPASS], rk[NPASS], rv[NPASS];
    // buffer-descriptor loads (non-temporal): rows past N and padded head columns arrive as zeros, no address arithmetic
    const BufTileLoader<TIN, NPASS, RPP, 2> qload(qb, prm.qs.sn, N, D, srow, scol), kload(kb, prm.ks.sn, N, D, srow, scol),
        vload(vb, prm.vs.sn, N, D, srow, scol);
    auto issue = [&](int c) {
        qload.load(c, rq);
        kload.load(c, rk);
        vload.load(c, rv);
    };
    bf16x8 ones;
#pragma unroll
    for (int i = 0; i < 8; ++i) ones[i] = (__bf16)(r == 0 ? 1.0f : 0.0f);

    const int nchunks = (N + C - 1) / C;
    const int c_begin = seg * prm.cps, c_end = min(nchunks, c_begin + prm.cps);
    f32x4 s2acc[MT];               // S2[16mt + 4q4 + reg][16w + r]
    f32x4 s1acc, ksacc;            // S1[16w + r] (row 0), ksum[16w + 4q4 + reg] (column 0)
    auto publish = [&]() {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int off = img_off<DP>(16 * w + r, 2 * mt + (q4 >> 1)) + ((q4 & 1) << 3);
            *reinterpret_cast<bf16x4*>(smem + S2I + off) = to_bf16x4(s2acc[mt] * a);
        }
        if (r == 0) {
            const int off = img_off<DP>(DP, 2 * w + (q4 >> 1)) + ((q4 & 1) << 3);
            *reinterpret_cast<bf16x4*>(smem + S2I + off) = to_bf16x4(ksacc * a);
        }
        if (q4 == 0) reinterpret_cast<float*>(smem + S1V)[16 * w + r] = s1acc[0];
    };
    for (int i = tid; i < SIMG / 16; i += 512) *reinterpret_cast<f32x4*>(smem + S2I + 16 * i) = f32x4{0, 0, 0, 0};
    if (tid < DP) reinterpret_cast<float*>(smem + S1V)[tid] = 0.f;
    s1acc = f32x4{0, 0, 0, 0};
    ksacc = f32x4{0, 0, 0, 0};
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) s2acc[mt] = f32x4{0, 0, 0, 0};
    if (seg > 0) {
        __syncthreads();
        const float* rec = prm.state + ((int64_t)bh * (prm.nseg - 1) + (seg - 1)) * (DP * DP + 2 * DP);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int i = 0; i < 4; ++i) s2acc[mt][i] = rec[(16 * mt + 4 * q4 + i) * DP + 16 * w + r];
        if (q4 == 0) s1acc[0] = rec[DP * DP + 16 * w + r];
        if (r == 0)
#pragma unroll
            for (int i = 0; i < 4; ++i) ksacc[i] = rec[DP * DP + DP + 16 * w + 4 * q4 + i];
        publish();
    }
    issue(c_begin);
    __syncthreads();

    for (int c = c_begin; c < c_end; ++c) {
        const int n0 = c * C;
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int row = srow + ps * RPP;
            if constexpr (NORM) {
                float xq[EPL], xk[EPL];
                piece_to_float<TIN>(rq[ps], xq);
                piece_to_float<TIN>(rk[ps], xk);
                float sq = 0.f, sk = 0.f;
#pragma unroll
                for (int e = 0; e < EPL; ++e) { sq += xq[e]; sk += xk[e]; }
                // (x - mean) * scale as one fma per element; a row past N was loaded as zeros (mean 0 -> 0), a padded head
                // column has its scale zeroed (qsc_c / ksc_c)
                const float nmq = -rowgroup_allsum_dpp<COLS>(sq) * invD * qsc_c, nmk = -rowgroup_allsum_dpp<COLS>(sk) * invD * ksc_c;
#pragma unroll
                for (int e = 0; e < EPL; ++e) {
                    xq[e] = fmaf(xq[e], qsc_c, nmq);
                    xk[e] = fmaf(xk[e], ksc_c, nmk);
                }
                stage_floats<DP, EPL, 1>(smem, QI, row, scol, xq);
                stage_floats<DP, EPL, 1>(smem, KI, row, scol, xk);
            } else {
                *reinterpret_cast<u32x4*>(smem + QI + img_off<DP>(row, scol)) = rq[ps];
                *reinterpret_cast<u32x4*>(smem + KI + img_off<DP>(row, scol)) = rk[ps];
            }
            *reinterpret_cast<u32x4*>(smem + VI + img_off<DP>(row, scol)) = rv[ps];
        }
        if (c + 1 < c_end) issue(c + 1);
        __syncthreads();                                             // B1
        // ---- phase A: query tile qt, output columns of d-half dh --------------------------------------------
        const int qi = 16 * qt + r;
        bf16x8 qf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = ld_row8<DP>(smem, QI, qi, 4 * ks + q4);
        f32x4 oacc[4];
        f32x4 qkacc = {0, 0, 0, 0};
#pragma unroll
        for (int t = 0; t <= 4; ++t) {                               // t == 4: the (a ksum) row tile
            const int dt = t < 4 ? 4 * dh + t : MT;
            f32x4 acc = {0, 0, 0, 0};
            if (t < 4) acc = *reinterpret_cast<const f32x4*>(smem + S1V + (16 * dt + 4 * q4) * 4);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                acc = mfma(*reinterpret_cast<const bf16x8*>(smem + S2I + img_off<DP>(16 * dt + r, 4 * ks + q4)), qf[ks], acc);
            }
            if (t < 4) oacc[t < 4 ? t : 0] = acc;
            else qkacc = acc;
        }
        const float qk = __shfl(qkacc[0], r, 64);
        float gsum = 0.f;
        Frag<2> pf[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            f32x4 pt[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int jt = 2 * s + e;
                f32x4 sc = {0, 0, 0, 0};
                if (jt <= qt) {
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) sc = mfma(ld_row8<DP>(smem, KI, 16 * jt + r, 4 * ks + q4), qf[ks], sc);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bool keep = (jt < qt) || (jt == qt && (4 * q4 + i) <= r);
                    const float sv = keep ? a * sc[i] : 0.f;
                    gsum += sv;
                    pt[e][i] = keep ? 1.0f + sv : 0.f;
                }
            }
            bf16x4 h0, l0, h1, l1;
            split4(pt[0], h0, l0);
            split4(pt[1], h1, l1);
            pf[s].p[0] = cat4(h0, h1);
            pf[s].p[1] = cat4(l0, l1);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (2 * s <= qt) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const bf16x8 vf = ld_tr8<DP>(smem, VI, 32 * s, 16 * (4 * dh + t), lane);
                    oacc[t] = mfma(vf, pf[s].p[0], oacc[t]);
                    oacc[t] = mfma(vf, pf[s].p[1], oacc[t]);
                }
            }
        }
        gsum += __shfl_xor(gsum, 16, 64);
        gsum += __shfl_xor(gsum, 32, 64);
        const int gi = n0 + qi;
        const float gval = (float)(gi + 1) + qk + gsum;
        const float ginv = 1.0f / gval;
        if (gi < N && prm.g && q4 == 0 && dh == 0) prm.g[(int64_t)bh * N + gi] = gval;
        {   // 16 rows x 64 bf16 columns, staged through this wave's private 2 KB and stored as 128-byte row pieces
            char* ost = smem + OST + w * 2048;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int c8 = 4 * t + q4;                           // 8-byte unit inside the 128-byte row piece
                *reinterpret_cast<bf16x4*>(ost + r * 128 + ((((c8 >> 1) ^ r) & 7) << 4) + ((c8 & 1) << 3)) = to_bf16x4(oacc[t] * ginv);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int idx = u * 64 + lane, rl = idx >> 3, cc = idx & 7;
                const u32x4 val = *reinterpret_cast<const u32x4*>(ost + rl * 128 + (((cc ^ rl) & 7) << 4));
                const int go = n0 + 16 * qt + rl, col = 64 * dh + 8 * cc;
                if (go < N && col < D)
                    __builtin_nontemporal_store(val, reinterpret_cast<u32x4*>(reinterpret_cast<uint16_t*>(prm.o) + ((int64_t)bh * N + go) * D + col));
            }
        }
        // ---- phase B: S2[:, 16w ..] += K^T V, S1 += 1^T V, ksum += K^T 1 ---------------------------------------
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const bf16x8 vf = ld_tr8<DP>(smem, VI, 32 * s, 16 * w, lane);
            s1acc = mfma(ones, vf, s1acc);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const bf16x8 kf = ld_tr8<DP>(smem, KI, 32 * s, 16 * mt, lane);
                s2acc[mt] = mfma(kf, vf, s2acc[mt]);
                if (mt == w) ksacc = mfma(kf, ones, ksacc);
            }
        }
        __syncthreads();                                             // B2
        if (c + 1 < c_end) publish();
    }
}

template <bool NORM>
static int launch_bf16_d128(const Bf16Params& prm, int nb, hipStream_t stream) {
    constexpr int DP = 128;
    constexpr int lds = 3 * 64 * DP * 2 + (DP + 16) * DP * 2 + DP * 4 + 8 * 2048;
    auto kern = fwd_p1_mfma_bf16_d128_kernel<NORM>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(nb), dim3(512), lds, stream, prm);
    return (int)hipGetLastError();
}

template <int DP, bool NORM, int LEAN>
static int launch_bf16_l(const Bf16Params& prm, int nb, hipStream_t stream) {
    constexpr int lds = 3 * 64 * DP * 2 + (LEAN ? 1 : 2) * (DP + 16) * DP * 2 + DP * 4;
    auto kern = fwd_p1_mfma_bf16_kernel<DP, NORM, LEAN>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(nb), dim3(256), lds, stream, prm);
    return (int)hipGetLastError();
}

template <int DP, bool NORM>
static int launch_bf16_t(const Bf16Params& prm, int nb, hipStream_t stream) {
    // default 2 = single-part state image, two-part P: measured error = the bf16 rounding of the result (1.578e-3 vs 1.572e-3
    // normwise), 8 % faster than 0 (both two-part); 1 = P single-part as well: 12 % faster, 1.4x the rounding error
    static const int lean = [] { const char* e = getenv("FASTMAX_BF16_LEAN"); return e ? atoi(e) : 2; }();
    if (lean == 2) return launch_bf16_l<DP, NORM, 2>(prm, nb, stream);
    return lean ? launch_bf16_l<DP, NORM, 1>(prm, nb, stream) : launch_bf16_l<DP, NORM, 0>(prm, nb, stream);
}

bool mfma_bf16_supported(const fastmax_problem& p) {
    return p.p == 1 && p.causal && p.in_dtype == FASTMAX_BF16 && p.out_dtype == FASTMAX_BF16 && (p.D % 8) == 0 && p.D <= 128;
}

int launch_fwd_mfma_bf16(const FwdArgs& a, const float* qscale, const float* kscale) {
    if (!mfma_bf16_supported(a.prob)) return FASTMAX_E_BAD_SHAPE;
    // 31-bit buffer offsets inside one (b,h) slab; anything larger goes to the generic kernel (64-bit addresses)
    if (!(quad32_span_ok(a.qs.sn, a.prob.Nq, a.prob.D, 2) && quad32_span_ok(a.ks.sn, a.prob.Nq, a.prob.D, 2) &&
          quad32_span_ok(a.vs.sn, a.prob.Nq, a.prob.D, 2)))
        return launch_fwd_mfma_gen(a, qscale, kscale);
    const SplitPlan plan = split_plan(a.prob);
    const int dp = a.prob.D <= 64 ? 64 : 128;
    if (plan.nseg > 1) {
        if (!a.workspace || a.workspace_bytes < split_workspace_bytes(a.prob, dp)) return FASTMAX_E_WORKSPACE;
        const int rc = a.stats ? linearmax_stats_and_states(a, plan, dp) : launch_split_states(a, plan, dp, kscale);
        if (rc) return rc;
    } else if (a.stats) {
        const int rc = linearmax_stats_and_states(a, plan, dp);
        if (rc) return rc;
    }
    Bf16Params prm{a.q, a.k, a.v, a.qs, a.ks, a.vs, a.o, a.g, qscale, kscale, a.prob.H, a.prob.Nq, a.prob.D, a.prob.a,
                   reinterpret_cast<const float*>(a.workspace), plan.nseg, plan.cps};
    const int nb = a.prob.B * a.prob.H * plan.nseg;
    const bool norm = qscale != nullptr;
    if (dp == 64) return norm ? launch_bf16_t<64, true>(prm, nb, a.stream) : launch_bf16_t<64, false>(prm, nb, a.stream);
    return norm ? launch_bf16_d128<true>(prm, nb, a.stream) : launch_bf16_d128<false>(prm, nb, a.stream);
}

}  // namespace fastmax
