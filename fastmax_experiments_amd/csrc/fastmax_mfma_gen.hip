// fastmax p=1 masked forward, linear in N, matrix cores -- the generic sibling of fastmax_mfma.hip:
// any input dtype (fp32 / fp16 as bf16 hi+lo parts, bf16 as exact single parts), head sizes padded to
// DP = 64 or 128 columns in LDS, optional fused linearmax prologue.
//
// Same algorithm as the D=64 fp32 kernel (see its header): per 64-token chunk
//   (3) O^T = S1 + (a S2)^T Q^T     (1) S^T = K Q^T, P = 1 + a s (masked)     (2) O^T += V^T P^T
//   g_i = (i+1) + a q_i.ksum_prev + a sum_{j<=i} s_ij         (4) S2 += K^T V
// with S2 (D x D fp32) carried in MFMA accumulators, S1 / ksum exact fp32 in LDS.  Differences: Q is kept
// unscaled in its image (bf16 inputs stay exact) and a = 1/nt is applied to the scores and folded into
// the S2 image; the staging maps are computed from the element size.
//
// Fused linearmax prologue (fastmax_hack.py:38-43): with per-(b,h) scales sq, sk (1 / max token norm,
// from fastmax_hip_normalize_stats) each staged Q / K row becomes (x - mean_D x) * scale before it is
// split into bf16 parts -- no normalised copy of Q, K ever goes to HBM.
#include "fastmax_mfma_common.h"

namespace fastmax {

struct GenParams {
    const void *q, *k, *v;
    Strides3 qs, ks, vs;
    void* o;
    float* g;
    const float *qscale, *kscale;     // (B*H) each or null: fused linearmax normalisation
    int H, N, D, out_dtype;
    float a;
    const float* state;   // sequence split: inclusive prefix states [(bh*(nseg-1) + seg-1)][DP*DP + 2*DP], or null
    int nseg, cps;
};

// grid = B*H, block = 256.  NORM: fused linearmax prologue on Q and K.
template <int DP, typename TIN, bool NORM>
__global__ __launch_bounds__(256, DP == 64 ? 2 : 1) void fwd_p1_mfma_gen_kernel(GenParams prm) {
    // bf16 inputs stay single-part even when normalised in-kernel: the reference itself forms (q - mean)/max
    // in the input dtype (fastmax_hack.py:38-43), so rounding the normalised row to bf16 is its own precision
    constexpr int NP = InTraits<TIN>::NP;
    constexpr int NPV = InTraits<TIN>::NP;
    constexpr int EPL = InTraits<TIN>::EPL;
    constexpr int C = 64, IMG = C * DP * 2, SIMG = DP * DP * 2;
    constexpr int QI = 0, KI = NP * IMG, VI = 2 * NP * IMG;
    constexpr int S2I = 2 * NP * IMG + NPV * IMG;               // a*S2^T image [d][m], hi at S2I, lo at S2I + SIMG
    constexpr int S1V = S2I + 2 * SIMG, KSUM = S1V + 2 * DP * 4;
    constexpr int COLS = DP / EPL, RPP = 256 / COLS, NPASS = C / RPP;
    constexpr int PARTV = KSUM + 2 * DP * 4, PARTK = PARTV + RPP * DP * 4, QK = PARTK + RPP * DP * 4;
    constexpr int KS = DP / 32, DT = DP / 16, MT = DP / 16;
    constexpr int NSL = DP / 64;                                 // S2 value-column tiles owned per wave
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

    u32x4 rq[NPASS], rk[NPASS], rv[NPASS];
    const TileLoader<TIN, NPASS, RPP, true> qload(qb, prm.qs.sn, N, D, DP, srow, scol), kload(kb, prm.ks.sn, N, D, DP, srow, scol),
        vload(vb, prm.vs.sn, N, D, DP, srow, scol);
    auto issue = [&](int n0) {
        qload.load(n0 / C, rq);
        kload.load(n0 / C, rk);
        vload.load(n0 / C, rv);
    };
    const int nchunks = (N + C - 1) / C;
    const int c_begin = seg * prm.cps, c_end = min(nchunks, c_begin + prm.cps);
    f32x4 s2acc[NSL][MT];                                       // S2[16mt + 4q4 + reg][16(w + 4sl) + r]
    auto publish_s2 = [&]() {                                   // a * accumulators -> bf16 hi/lo image rows d
#pragma unroll
        for (int sl = 0; sl < NSL; ++sl)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                bf16x4 hi, lo;
                split4(s2acc[sl][mt] * a, hi, lo);
                const int off = img_off<DP>(16 * (w + 4 * sl) + r, 2 * mt + (q4 >> 1)) + ((q4 & 1) << 3);
                *reinterpret_cast<bf16x4*>(smem + S2I + off) = hi;
                *reinterpret_cast<bf16x4*>(smem + S2I + SIMG + off) = lo;
            }
    };
    if (seg == 0) {
        for (int i = tid; i < (2 * SIMG) / 16; i += 256) *reinterpret_cast<f32x4*>(smem + S2I + 16 * i) = f32x4{0, 0, 0, 0};
        if (tid < DP) {
            reinterpret_cast<float*>(smem + S1V)[tid] = 0.f;
            reinterpret_cast<float*>(smem + KSUM)[tid] = 0.f;
        }
#pragma unroll
        for (int sl = 0; sl < NSL; ++sl)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) s2acc[sl][mt] = f32x4{0, 0, 0, 0};
    } else {
        const float* rec = prm.state + ((int64_t)bh * (prm.nseg - 1) + (seg - 1)) * (DP * DP + 2 * DP);
#pragma unroll
        for (int sl = 0; sl < NSL; ++sl)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int i = 0; i < 4; ++i) s2acc[sl][mt][i] = rec[(16 * mt + 4 * q4 + i) * DP + 16 * (w + 4 * sl) + r];
        publish_s2();
        if (tid < DP) {
            reinterpret_cast<float*>(smem + S1V)[DP * (c_begin & 1) + tid] = rec[DP * DP + tid];
            reinterpret_cast<float*>(smem + KSUM)[DP * (c_begin & 1) + tid] = rec[DP * DP + DP + tid];
        }
    }
    issue(c_begin * C);
    __syncthreads();
    for (int c = c_begin; c < c_end; ++c) {
        const int n0 = c * C, cur = c & 1, nxt = cur ^ 1;
        const float* ksum_cur = reinterpret_cast<const float*>(smem + KSUM) + DP * cur;
        const float* s1v_cur = reinterpret_cast<const float*>(smem + S1V) + DP * cur;
        // ---- (a) registers -> bf16 images, exact fp32 side sums ------------------------------------
        {
            float ksv[EPL], ck[EPL], cv[EPL];
#pragma unroll
            for (int e = 0; e < EPL; ++e) { ksv[e] = ksum_cur[scol * EPL + e]; ck[e] = 0.f; cv[e] = 0.f; }
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps) {
                const int row = srow + ps * RPP;
                const bool rowok = n0 + row < N;
                float xq[EPL], xk[EPL], xv[EPL];
                piece_to_float<TIN>(rq[ps], xq);
                piece_to_float<TIN>(rk[ps], xk);
                piece_to_float<TIN>(rv[ps], xv);
                if constexpr (NORM) {
                    float sq = 0.f, sk = 0.f;
#pragma unroll
                    for (int e = 0; e < EPL; ++e) { sq += xq[e]; sk += xk[e]; }
                    const float mq = rowgroup_allsum<COLS>(sq) * invD, mk = rowgroup_allsum<COLS>(sk) * invD;
                    const bool live = colok && rowok;
#pragma unroll
                    for (int e = 0; e < EPL; ++e) {
                        xq[e] = live ? (xq[e] - mq) * qsc : 0.f;
                        xk[e] = live ? (xk[e] - mk) * ksc : 0.f;
                    }
                    stage_floats<DP, EPL, NP>(smem, QI, row, scol, xq);
                    stage_floats<DP, EPL, NP>(smem, KI, row, scol, xk);
                } else {
                    stage_piece<DP, TIN>(smem, QI, row, scol, rq[ps]);
                    stage_piece<DP, TIN>(smem, KI, row, scol, rk[ps]);
                }
                stage_piece<DP, TIN>(smem, VI, row, scol, rv[ps]);
                float part = 0.f;
#pragma unroll
                for (int e = 0; e < EPL; ++e) { part = fmaf(xq[e], ksv[e], part); ck[e] += xk[e]; cv[e] += xv[e]; }
                part = rowgroup_sum<COLS>(part);
                if (scol == COLS - 1) reinterpret_cast<float*>(smem + QK)[row] = a * part;
            }
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                reinterpret_cast<float*>(smem + PARTK)[srow * DP + scol * EPL + e] = ck[e];
                reinterpret_cast<float*>(smem + PARTV)[srow * DP + scol * EPL + e] = cv[e];
            }
        }
        if (c + 1 < c_end) issue(n0 + C);
        __syncthreads();                                             // B1
        for (int t = tid; t < 2 * DP; t += 256) {                    // running sums for the next chunk
            const int col = t % DP;
            const float* part = reinterpret_cast<const float*>(smem + (t < DP ? PARTV : PARTK));
            float* base = reinterpret_cast<float*>(smem + (t < DP ? S1V : KSUM));
            float s = base[DP * cur + col];
#pragma unroll 8
            for (int g16 = 0; g16 < RPP; ++g16) s += part[g16 * DP + col];
            base[DP * nxt + col] = s;
        }
        // ---- phase A ---------------------------------------------------------------------------------
        const int qi = 16 * w + r;
        Frag<NP> qf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int p = 0; p < NP; ++p) qf[ks].p[p] = ld_row8<DP>(smem, QI + p * IMG, qi, 4 * ks + q4);
        f32x4 oacc[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            oacc[dt] = *reinterpret_cast<const f32x4*>(s1v_cur + 16 * dt + 4 * q4);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                Frag<2> sf;                                          // rows d of the (a S2)^T image: 2*DP-byte rows
                sf.p[0] = *reinterpret_cast<const bf16x8*>(smem + S2I + img_off<DP>(16 * dt + r, 4 * ks + q4));
                sf.p[1] = *reinterpret_cast<const bf16x8*>(smem + S2I + SIMG + img_off<DP>(16 * dt + r, 4 * ks + q4));
                oacc[dt] = mfma_parts<2, NP>(sf, qf[ks], oacc[dt]);
            }
        }
        float gsum = 0.f;
        Frag<2> pf[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            f32x4 pt[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int jt = 2 * s + e;
                f32x4 sc = {0, 0, 0, 0};
                if (jt <= w) {
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        Frag<NP> kf;
#pragma unroll
                        for (int p = 0; p < NP; ++p) kf.p[p] = ld_row8<DP>(smem, KI + p * IMG, 16 * jt + r, 4 * ks + q4);
                        sc = mfma_parts<NP, NP>(kf, qf[ks], sc);
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bool keep = (jt < w) || (jt == w && (4 * q4 + i) <= r);
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
            if (2 * s <= w) {
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    Frag<NPV> vf;
#pragma unroll
                    for (int p = 0; p < NPV; ++p) vf.p[p] = ld_tr8<DP>(smem, VI + p * IMG, 32 * s, 16 * dt, lane);
                    oacc[dt] = mfma_parts<NPV, 2>(vf, pf[s], oacc[dt]);
                }
            }
        }
        gsum += __shfl_xor(gsum, 16, 64);
        gsum += __shfl_xor(gsum, 32, 64);
        const int gi = n0 + qi;
        const float gval = (float)(gi + 1) + reinterpret_cast<const float*>(smem + QK)[qi] + gsum;
        const float ginv = 1.0f / gval;
        if (gi < N && prm.g && q4 == 0) prm.g[(int64_t)bh * N + gi] = gval;
        // output rows, staged in the result dtype through this wave's own (already consumed) Q image rows
        store_tile16_private<DP, sizeof(TIN)>(smem + QI + 16 * w * (2 * DP), smem + QI + IMG + 16 * w * (2 * DP), oacc, ginv,
                                              lane, prm.o, prm.out_dtype, ((int64_t)bh * N + n0 + 16 * w) * D, n0 + 16 * w, N, D);
        // ---- phase B: S2[:, 16(w+4sl) ..] += K^T V ------------------------------------------------------
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int sl = 0; sl < NSL; ++sl) {
                Frag<NPV> vf;
#pragma unroll
                for (int p = 0; p < NPV; ++p) vf.p[p] = ld_tr8<DP>(smem, VI + p * IMG, 32 * s, 16 * (w + 4 * sl), lane);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    Frag<NP> kf;
#pragma unroll
                    for (int p = 0; p < NP; ++p) kf.p[p] = ld_tr8<DP>(smem, KI + p * IMG, 32 * s, 16 * mt, lane);
                    s2acc[sl][mt] = mfma_parts<NP, NPV>(kf, vf, s2acc[sl][mt]);
                }
            }
        }
        __syncthreads();                                             // B2: all reads of this chunk's images done
        if (c + 1 < c_end) publish_s2();
    }
}

template <int DP, typename TIN> constexpr int gen_lds_bytes(bool norm) {
    const int NP = InTraits<TIN>::NP, NPV = InTraits<TIN>::NP, EPL = InTraits<TIN>::EPL;
    (void)norm;
    const int IMG = 64 * DP * 2, SIMG = DP * DP * 2, RPP = 256 / (DP / EPL);
    return 2 * NP * IMG + NPV * IMG + 2 * SIMG + 4 * DP * 4 + 2 * RPP * DP * 4 + 256;
}

template <int DP, typename TIN, bool NORM>
static int launch_gen_t(const GenParams& prm, int nb, hipStream_t stream) {
    constexpr int lds = gen_lds_bytes<DP, TIN>(NORM);
    static_assert(lds <= 160 * 1024, "LDS budget");
    auto kern = fwd_p1_mfma_gen_kernel<DP, TIN, NORM>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(nb), dim3(256), lds, stream, prm);
    return (int)hipGetLastError();
}
template <typename TIN, bool NORM>
static int launch_gen_d(const GenParams& prm, int nb, hipStream_t stream) {
    if (prm.D <= 64) return launch_gen_t<64, TIN, NORM>(prm, nb, stream);
    if constexpr (InTraits<TIN>::NP == 2) return FASTMAX_E_BAD_SHAPE;            // 128-wide split images exceed the LDS
    else return launch_gen_t<128, TIN, NORM>(prm, nb, stream);
}

bool mfma_gen_supported(const fastmax_problem& p, bool norm) {
    if (!(p.p == 1 && p.causal) || p.in_dtype != p.out_dtype) return false;
    const int epl = p.in_dtype == FASTMAX_F32 ? 4 : 8;
    if (p.D % epl) return false;
    if (p.D <= 64) return true;
    (void)norm;
    return p.D <= 128 && p.in_dtype == FASTMAX_BF16;
}

int launch_fwd_mfma_gen(const FwdArgs& a, const float* qscale, const float* kscale) {
    const bool norm = qscale != nullptr;
    if (!mfma_gen_supported(a.prob, norm)) return FASTMAX_E_BAD_SHAPE;
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
    GenParams prm{a.q, a.k, a.v, a.qs, a.ks, a.vs, a.o, a.g, qscale, kscale, a.prob.H, a.prob.Nq, a.prob.D,
                  a.prob.out_dtype, a.prob.a, reinterpret_cast<const float*>(a.workspace), plan.nseg, plan.cps};
    const int nb = a.prob.B * a.prob.H * plan.nseg;
    switch (a.prob.in_dtype) {
        case FASTMAX_F32: return norm ? launch_gen_d<float, true>(prm, nb, a.stream) : launch_gen_d<float, false>(prm, nb, a.stream);
        case FASTMAX_BF16: return norm ? launch_gen_d<bf16_t, true>(prm, nb, a.stream) : launch_gen_d<bf16_t, false>(prm, nb, a.stream);
        case FASTMAX_F16: return norm ? launch_gen_d<f16_t, true>(prm, nb, a.stream) : launch_gen_d<f16_t, false>(prm, nb, a.stream);
    }
    return FASTMAX_E_BAD_DTYPE;
}

}  // namespace fastmax
