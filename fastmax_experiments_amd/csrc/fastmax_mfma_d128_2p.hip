// fastmax p=1 masked forward, linear in N, for fp32 / fp16 tensors at 64 < D <= 128 (two bf16 parts per operand).
//
// Same chunked scan as fastmax_mfma_bf16.hip's eight-wave D = 128 kernel (wave (qt, dh): query tile qt = w & 3, output
// columns [64 dh, 64 dh + 64); the 128 x 128 state spread over the eight waves), with hi + lo operand parts
// (hi.hi + lo.hi + hi.lo, ~2^-16 relative).  The hi / lo images of Q, K, V and of the state would need 168 KB of LDS,
// so Q never gets an image: a wave's own 16 query rows are loaded from global memory straight into B fragments (one chunk
// ahead, raw, split in registers), and the result tile is stored from the accumulators (16-byte pieces per lane) instead
// of through a staging area.  LDS: K, V images 64 KB + state images 72 KB + S1.
// NORM fuses the linearmax prologue (fastmax_hack.py:38-43): K rows in staging, Q rows in registers.
#include "fastmax_mfma_common.h"

namespace fastmax {

struct D128Params {
    const void *q, *k, *v;
    Strides3 qs, ks, vs;
    void* o;
    float* g;
    const float *qscale, *kscale;
    int H, N, D, out_dtype;
    float a;
    const float* state;
    int nseg, cps;
};

template <typename TIN, bool NORM, bool BUF>
__global__ __launch_bounds__(512, 1) void fwd_p1_mfma_d128_2p_kernel(D128Params prm) {
    constexpr int NP = 2, EPL = InTraits<TIN>::EPL;
    static_assert(InTraits<TIN>::NP == 2, "two-part operands");
    constexpr int DP = 128, C = 64, IMG = C * DP * 2, SIMG = (DP + 16) * DP * 2;
    constexpr int KI = 0, VI = NP * IMG, S2I = 2 * NP * IMG, S1V = S2I + 2 * SIMG;
    constexpr int COLS = DP / EPL, RPP = 512 / COLS, NPASS = C / RPP;
    constexpr int KS = DP / 32, MT = DP / 16, QL = 8 / EPL;                  // QL 16-byte loads per 8-element Q fragment piece
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

    u32x4 rk[NPASS], rv[NPASS], rq[KS][QL];
    const ScanLoader<BUF, TIN, NPASS, RPP> kload(kb, prm.ks.sn, N, D, DP, srow, scol), vload(vb, prm.vs.sn, N, D, DP, srow, scol);
    const RowPieceLoader<BUF, TIN> qrows(qb, prm.qs.sn, N, D);
    auto issue = [&](int c) {
        kload.load(c, rk);
        vload.load(c, rv);
        const int row = c * C + 16 * qt + r;                       // this wave's query row on this lane
        int q4o = q4;                                              // opaque: the eight piece offsets are re-formed per chunk, not hoisted and spilled
        asm volatile("" : "+v"(q4o));
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int u = 0; u < QL; ++u) rq[ks][u] = qrows.load(row, (32 * ks + 8 * q4o) / EPL + u);
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
            bf16x4 hi, lo;
            split4(s2acc[mt] * a, hi, lo);
            const int off = img_off<DP>(16 * w + r, 2 * mt + (q4 >> 1)) + ((q4 & 1) << 3);
            *reinterpret_cast<bf16x4*>(smem + S2I + off) = hi;
            *reinterpret_cast<bf16x4*>(smem + S2I + SIMG + off) = lo;
        }
        if (r == 0) {
            bf16x4 hi, lo;
            split4(ksacc * a, hi, lo);
            const int off = img_off<DP>(DP, 2 * w + (q4 >> 1)) + ((q4 & 1) << 3);
            *reinterpret_cast<bf16x4*>(smem + S2I + off) = hi;
            *reinterpret_cast<bf16x4*>(smem + S2I + SIMG + off) = lo;
        }
        if (q4 == 0) reinterpret_cast<float*>(smem + S1V)[16 * w + r] = s1acc[0];
    };
    for (int i = tid; i < (2 * SIMG) / 16; i += 512) *reinterpret_cast<f32x4*>(smem + S2I + 16 * i) = f32x4{0, 0, 0, 0};
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
        // ---- staging: K (optionally normalised) and V as hi / lo images; Q fragments in registers ----------------------
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int row = srow + ps * RPP;
            if constexpr (NORM) {
                float xk[EPL];
                piece_to_float<TIN>(rk[ps], xk);
                float sk = 0.f;
#pragma unroll
                for (int e = 0; e < EPL; ++e) sk += xk[e];
                const float mk = rowgroup_allsum<COLS>(sk) * invD;
                const bool live = colok && (n0 + row < N);
#pragma unroll
                for (int e = 0; e < EPL; ++e) xk[e] = live ? (xk[e] - mk) * ksc : 0.f;
                stage_floats<DP, EPL, NP>(smem, KI, row, scol, xk);
            } else {
                stage_piece<DP, TIN>(smem, KI, row, scol, rk[ps]);
            }
            stage_piece<DP, TIN>(smem, VI, row, scol, rv[ps]);
        }
        Frag<NP> qf[KS];
        {
            float xq[KS][8];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if constexpr (QL == 2) {
                    float lo4[4], hi4[4];
                    piece_to_float<TIN>(rq[ks][0], lo4);
                    piece_to_float<TIN>(rq[ks][1], hi4);
#pragma unroll
                    for (int i = 0; i < 4; ++i) { xq[ks][i] = lo4[i]; xq[ks][4 + i] = hi4[i]; }
                } else {
                    piece_to_float<TIN>(rq[ks][0], xq[ks]);
                }
            }
            if constexpr (NORM) {
                float sq = 0.f;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int i = 0; i < 8; ++i) sq += xq[ks][i];
                sq += __shfl_xor(sq, 16, 64);                       // the row is spread over the four q4 lanes
                sq += __shfl_xor(sq, 32, 64);
                const float mq = sq * invD;
                const bool rowlive = (n0 + 16 * qt + r) < N;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const bool live = rowlive && (32 * ks + 8 * q4 + i) < D;
                        xq[ks][i] = live ? (xq[ks][i] - mq) * qsc : 0.f;
                    }
            }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                bf16x4 h0, l0, h1, l1;
                split4(f32x4{xq[ks][0], xq[ks][1], xq[ks][2], xq[ks][3]}, h0, l0);
                split4(f32x4{xq[ks][4], xq[ks][5], xq[ks][6], xq[ks][7]}, h1, l1);
                qf[ks].p[0] = cat4(h0, h1);
                qf[ks].p[1] = cat4(l0, l1);
            }
        }
        if (c + 1 < c_end) issue(c + 1);
        __syncthreads();                                             // B1
        // ---- phase A: query tile qt, output columns of d-half dh ---------------------------------------------------------
        const int qi = 16 * qt + r;
        f32x4 oacc[4];
        f32x4 qkacc = {0, 0, 0, 0};
#pragma unroll
        for (int t = 0; t <= 4; ++t) {                               // t == 4: the (a ksum) row tile
            const int dt = t < 4 ? 4 * dh + t : MT;
            f32x4 acc = {0, 0, 0, 0};
            if (t < 4) acc = *reinterpret_cast<const f32x4*>(smem + S1V + (16 * dt + 4 * q4) * 4);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                Frag<2> sf;
                sf.p[0] = *reinterpret_cast<const bf16x8*>(smem + S2I + img_off<DP>(16 * dt + r, 4 * ks + q4));
                sf.p[1] = *reinterpret_cast<const bf16x8*>(smem + S2I + SIMG + img_off<DP>(16 * dt + r, 4 * ks + q4));
                acc = mfma_parts<2, 2>(sf, qf[ks], acc);
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
                    for (int ks = 0; ks < KS; ++ks) {
                        Frag<2> kf;
#pragma unroll
                        for (int p = 0; p < NP; ++p) kf.p[p] = ld_row8<DP>(smem, KI + p * IMG, 16 * jt + r, 4 * ks + q4);
                        sc = mfma_parts<2, 2>(kf, qf[ks], sc);
                    }
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
                    Frag<2> vf;
#pragma unroll
                    for (int p = 0; p < NP; ++p) vf.p[p] = ld_tr8<DP>(smem, VI + p * IMG, 32 * s, 16 * (4 * dh + t), lane);
                    oacc[t] = mfma_parts<2, 2>(vf, pf[s], oacc[t]);
                }
            }
        }
        gsum += __shfl_xor(gsum, 16, 64);
        gsum += __shfl_xor(gsum, 32, 64);
        const int gi = n0 + qi;
        const float gval = (float)(gi + 1) + qk + gsum;
        const float ginv = 1.0f / gval;
        if (gi < N && prm.g && q4 == 0 && dh == 0) prm.g[(int64_t)bh * N + gi] = gval;
        if (gi < N) {                                                // lane: row gi, columns 64 dh + 16 t + 4 q4 .. + 3
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int col = 64 * dh + 16 * t + 4 * q4;
                if (col < D) store4_any(prm.o, prm.out_dtype, ((int64_t)bh * N + gi) * D + col, oacc[t] * ginv);
            }
        }
        // ---- phase B: S2[:, 16w ..] += K^T V, S1 += 1^T V, ksum += K^T 1 (hi and lo parts of the ones products) ---------
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            Frag<2> vf;
#pragma unroll
            for (int p = 0; p < NP; ++p) vf.p[p] = ld_tr8<DP>(smem, VI + p * IMG, 32 * s, 16 * w, lane);
            s1acc = mfma(ones, vf.p[0], s1acc);
            s1acc = mfma(ones, vf.p[1], s1acc);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                Frag<2> kf;
#pragma unroll
                for (int p = 0; p < NP; ++p) kf.p[p] = ld_tr8<DP>(smem, KI + p * IMG, 32 * s, 16 * mt, lane);
                s2acc[mt] = mfma_parts<2, 2>(kf, vf, s2acc[mt]);
                if (mt == w) {
                    ksacc = mfma(kf.p[0], ones, ksacc);
                    ksacc = mfma(kf.p[1], ones, ksacc);
                }
            }
        }
        __syncthreads();                                             // B2
        if (c + 1 < c_end) publish();
    }
}

template <typename TIN, bool NORM, bool BUF>
static int launch_d128_2p_b(const D128Params& prm, int nb, hipStream_t stream) {
    constexpr int DP = 128;
    constexpr int lds = 4 * 64 * DP * 2 + 2 * (DP + 16) * DP * 2 + DP * 4;
    static_assert(lds <= 160 * 1024, "LDS budget");
    auto kern = fwd_p1_mfma_d128_2p_kernel<TIN, NORM, BUF>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(nb), dim3(512), lds, stream, prm);
    return (int)hipGetLastError();
}

template <typename TIN, bool NORM>
static int launch_d128_2p_t(const D128Params& prm, int nb, hipStream_t stream) {
    // K / V tiles through buffer descriptors when a (b,h) slab fits their 31-bit offsets
    const bool buf = quad32_span_ok(prm.ks.sn, prm.N, prm.D, (int)sizeof(TIN)) && quad32_span_ok(prm.vs.sn, prm.N, prm.D, (int)sizeof(TIN)) &&
                     quad32_span_ok(prm.qs.sn, prm.N, prm.D, (int)sizeof(TIN));
    return buf ? launch_d128_2p_b<TIN, NORM, true>(prm, nb, stream) : launch_d128_2p_b<TIN, NORM, false>(prm, nb, stream);
}

bool mfma_d128_2p_supported(const fastmax_problem& p) {
    if (!(p.p == 1 && p.causal) || p.D <= 64 || p.D > 128) return false;
    if (p.in_dtype == FASTMAX_F32) return p.out_dtype == FASTMAX_F32 && (p.D % 4) == 0;
    if (p.in_dtype == FASTMAX_F16) return p.out_dtype == FASTMAX_F16 && (p.D % 8) == 0;
    return false;
}

int launch_fwd_mfma_d128_2p(const FwdArgs& a, const float* qscale, const float* kscale) {
    if (!mfma_d128_2p_supported(a.prob)) return FASTMAX_E_BAD_SHAPE;
    const SplitPlan plan = split_plan(a.prob);
    if (plan.nseg > 1) {
        if (!a.workspace || a.workspace_bytes < split_workspace_bytes(a.prob, 128)) return FASTMAX_E_WORKSPACE;
        const int rc = a.stats ? linearmax_stats_and_states(a, plan, 128) : launch_split_states(a, plan, 128, kscale);
        if (rc) return rc;
    } else if (a.stats) {
        const int rc = linearmax_stats_and_states(a, plan, 128);
        if (rc) return rc;
    }
    D128Params prm{a.q, a.k, a.v, a.qs, a.ks, a.vs, a.o, a.g, qscale, kscale, a.prob.H, a.prob.Nq, a.prob.D, a.prob.out_dtype,
                   a.prob.a, reinterpret_cast<const float*>(a.workspace), plan.nseg, plan.cps};
    const int nb = a.prob.B * a.prob.H * plan.nseg;
    const bool norm = qscale != nullptr;
    if (a.prob.in_dtype == FASTMAX_F32)
        return norm ? launch_d128_2p_t<float, true>(prm, nb, a.stream) : launch_d128_2p_t<float, false>(prm, nb, a.stream);
    return norm ? launch_d128_2p_t<f16_t, true>(prm, nb, a.stream) : launch_d128_2p_t<f16_t, false>(prm, nb, a.stream);
}

}  // namespace fastmax
