// fastmax forward as masked / unmasked polynomial attention tiles on the matrix cores (gfx950):
//     o_i = sum_j f(a q_i.k_j) v_j / g_i ,  g_i = sum_j f(a q_i.k_j)   (j <= i when causal)
// f(s) = 1+s (p=1) or 1+s+s^2/2 (p=2).  Same function as the reference's factorised sums
// (attention_mechanisms/fastmax.py:184-322); for p=2 the factorised form costs ~4 D^2 (D+1) flop per
// token against ~4 D N/2 here, so below N ~ 2 D^2 (every configuration in BASELINE.json) this is the
// cheaper evaluation, and it needs no D^2 x (D+1) third-order state.  Also serves the unmasked and
// N_q != N_k (KV-cache decode) cases for both p.
//
// One workgroup = 64 queries of one (b,h) head (wave w: queries 16w..16w+15), looping over 64-key tiles:
//   (1) S^T = K Q^T        MFMA 16x16x32 bf16, A = K rows (ds_read_b128), B = Q rows
//   P = f(a S) * mask, split hi/lo in registers (accumulator layout = next B operand, permuted k)
//   (2) O^T += V^T P^T     A = V^T by ds_read_b64_tr_b16 of the row-major V image
// fp32 / fp16 inputs are carried as bf16 hi + lo parts (3-term products, ~2^-16 relative); bf16 inputs
// are exact single parts.  Head sizes are padded to DP = 64, 128 or 256 columns inside LDS only.
#include "fastmax_mfma_common.h"

#include <cstdlib>
#include <type_traits>

namespace fastmax {

struct QuadMfmaParams {
    const void *q, *k, *v;
    Strides3 qs, ks, vs;
    void* o;
    float* g;
    int H, Nq, Nk, D, causal, out_dtype;
    float a, g0;
};

// QG = 16-query groups per wave: a workgroup covers 64*QG queries, so every K / V fragment read from LDS feeds QG
// MFMAs and the per-tile barriers and staging are amortised over QG times the work.  Measured on MI355X (round 1):
// QG = 2 costs more in occupancy (registers 117 -> 148..254, LDS +50 %) than the reuse returns at N = 4096
// (p=2 bf16 2.23 -> 2.58 ms, fp32 5.47 -> 6.32 ms; only N = 2048 gained, 0.40 -> 0.36 ms), so QG = 1 ships.
template <int DP, typename TIN, int NPP> constexpr int quad_qg() { return 1; }

// grid = (ceil(Nq/(64 QG)), B*H), block = 256, dynamic LDS = (QG + 2) * NP * 64*DP*2 bytes.  DP = 256 (head sizes 136 .. 256,
// e.g. pythia-1b, Gemma: lit_gpt/config.py): the Q image is only read into registers before the first key tile, so it shares
// its LDS with the K / V images (2 * NP * 64*DP*2 bytes: 64 KB bf16, 128 KB two-part); one workgroup per CU.
template <int DP, int P, typename TIN, int NPP, int PFD>
__global__ __launch_bounds__(256, (DP == 64 || (DP == 128 && InTraits<TIN>::NP == 1)) ? 2 : 1) void fwd_quad_mfma_kernel(QuadMfmaParams prm) {
    constexpr int NP = InTraits<TIN>::NP, EPL = InTraits<TIN>::EPL;
    constexpr int QG = quad_qg<DP, TIN, NPP>(), QT = 64 * QG;
    constexpr int IMG = 64 * DP * 2, QIMG = QG * IMG;
    constexpr int QI = 0, KI = DP > 128 ? 0 : NP * QIMG, VI = KI + NP * IMG;
    static_assert(DP <= 128 || QG == 1, "shared Q / K image");
    constexpr int COLS = DP / EPL, RPP = 256 / COLS, NPASS = 64 / RPP;
    constexpr int KS = DP / 32, DT = DP / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q4 = lane >> 4;
    const int bh = blockIdx.y, b = bh / prm.H, h = bh % prm.H;
    const int D = prm.D, Nq = prm.Nq, Nk = prm.Nk;
    const bool causal = prm.causal != 0;
    const int nqt = gridDim.x;
    const int qt = causal ? nqt - 1 - (int)blockIdx.x : (int)blockIdx.x;      // causal: heaviest query tiles first
    const int i0 = qt * QT;
    const TIN* qb = reinterpret_cast<const TIN*>(prm.q) + (int64_t)b * prm.qs.sb + (int64_t)h * prm.qs.sh;
    const TIN* kb = reinterpret_cast<const TIN*>(prm.k) + (int64_t)b * prm.ks.sb + (int64_t)h * prm.ks.sh;
    const TIN* vb = reinterpret_cast<const TIN*>(prm.v) + (int64_t)b * prm.vs.sb + (int64_t)h * prm.vs.sh;
    const int srow = tid / COLS, scol = tid % COLS;

    // Q tile -> images (once); part p of the Q image lives at QI + p*QIMG
#pragma unroll
    for (int ps = 0; ps < NPASS * QG; ++ps) {
        const int row = srow + ps * RPP;
        const u32x4 raw = load_piece<TIN>(qb, prm.qs.sn, i0 + row, Nq, scol, D);
        if constexpr (NP == 1) {
            *reinterpret_cast<u32x4*>(smem + QI + img_off<DP>(row, scol)) = raw;
        } else {
            float x[EPL];
            piece_to_float<TIN>(raw, x);
#pragma unroll
            for (int hseg = 0; hseg < EPL / 4; ++hseg) {
                const f32x4 v4 = {x[4 * hseg], x[4 * hseg + 1], x[4 * hseg + 2], x[4 * hseg + 3]};
                bf16x4 hi, lo;
                split4(v4, hi, lo);
                const int e0 = scol * EPL + 4 * hseg, off = img_off<DP>(row, e0 >> 3) + ((e0 & 7) << 1);
                *reinterpret_cast<bf16x4*>(smem + QI + off) = hi;
                *reinterpret_cast<bf16x4*>(smem + QI + QIMG + off) = lo;
            }
        }
    }
    // PFD key tiles are in flight in registers (K and V of tile kt + PFD are requested while tile kt is staged)
    u32x4 rk[PFD][NPASS], rv[PFD][NPASS];
    const TileLoader<TIN, NPASS, RPP> kload(kb, prm.ks.sn, Nk, D, DP, srow, scol), vload(vb, prm.vs.sn, Nk, D, DP, srow, scol);
    const int nkt = causal ? min((i0 + QT + 63) / 64, (Nk + 63) / 64) : (Nk + 63) / 64;
#pragma unroll
    for (int f = 0; f < PFD; ++f)
        if (f < nkt) {
            kload.load(f, rk[f]);
            vload.load(f, rv[f]);
        }
    __syncthreads();
    Frag<NP> qf[QG][KS];
    int qidx[QG];
#pragma unroll
    for (int g = 0; g < QG; ++g) {
        qidx[g] = i0 + 16 * (QG * w + g) + r;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int p = 0; p < NP; ++p) qf[g][ks].p[p] = ld_row8<DP>(smem, QI + p * QIMG, 16 * (QG * w + g) + r, 4 * ks + q4);
    }
    f32x4 oacc[QG][DT];
    float gsum[QG];
#pragma unroll
    for (int g = 0; g < QG; ++g) {
        gsum[g] = 0.f;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) oacc[g][dt] = f32x4{0, 0, 0, 0};
    }
    const float a = prm.a;

    // One key tile.  MASKED = a tile that touches the causal diagonal or runs past N_k: only those pay for the
    // per-element compares.  NPP: parts of P (a bf16 problem with a bf16 result carries P as one rounded part: its
    // rounding is of the size of the output rounding itself; everything else keeps hi + lo).
    auto tile = [&](int kt, auto masked_tag) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        Frag<NPP> pf[QG][2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            f32x4 pt[QG][2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int jt = 2 * s + e;
                f32x4 sc[QG];
#pragma unroll
                for (int g = 0; g < QG; ++g) sc[g] = f32x4{0, 0, 0, 0};
                // sub-tile entirely above the diagonal for every query group of this wave (wave-uniform)
                const bool skip = MASKED && causal && (kt * 64 + 16 * jt) > (i0 + 16 * (QG * w + QG - 1) + 15);
                if (!skip) {
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        Frag<NP> kf;
#pragma unroll
                        for (int p = 0; p < NP; ++p) kf.p[p] = ld_row8<DP>(smem, KI + p * IMG, 16 * jt + r, 4 * ks + q4);
#pragma unroll
                        for (int g = 0; g < QG; ++g) sc[g] = mfma_parts<NP, NP>(kf, qf[g][ks], sc[g]);
                    }
                }
#pragma unroll
                for (int g = 0; g < QG; ++g)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float pv = poly_f<P>(a * sc[g][i]);
                        if constexpr (MASKED) {
                            const int key = kt * 64 + 16 * jt + 4 * q4 + i;
                            const bool keep = key < Nk && (!causal || key <= qidx[g]);
                            pv = keep ? pv : 0.f;
                        }
                        gsum[g] += pv;
                        pt[g][e][i] = pv;
                    }
            }
#pragma unroll
            for (int g = 0; g < QG; ++g) {
                if constexpr (NPP == 2) {
                    bf16x4 h0, l0, h1, l1;
                    split4(pt[g][0], h0, l0);
                    split4(pt[g][1], h1, l1);
                    pf[g][s].p[0] = cat4(h0, h1);
                    pf[g][s].p[1] = cat4(l0, l1);
                } else {
                    pf[g][s].p[0] = cat4(to_bf16x4(pt[g][0]), to_bf16x4(pt[g][1]));
                }
            }
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const bool skip = MASKED && causal && (kt * 64 + 32 * s) > (i0 + 16 * (QG * w + QG - 1) + 15);
            if (!skip) {
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    Frag<NP> vf;
#pragma unroll
                    for (int p = 0; p < NP; ++p) vf.p[p] = ld_tr8<DP>(smem, VI + p * IMG, 32 * s, 16 * dt, lane);
#pragma unroll
                    for (int g = 0; g < QG; ++g) oacc[g][dt] = mfma_parts<NP, NPP>(vf, pf[g][s], oacc[g][dt]);
                }
            }
        }
    };
    for (int kt0 = 0; kt0 < nkt; kt0 += PFD) {
#pragma unroll
        for (int f = 0; f < PFD; ++f) {
            const int kt = kt0 + f;
            if (kt >= nkt) break;
            __syncthreads();                               // previous tile fully consumed
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps) {
                stage_piece<DP, TIN>(smem, KI, srow + ps * RPP, scol, rk[f][ps]);
                stage_piece<DP, TIN>(smem, VI, srow + ps * RPP, scol, rv[f][ps]);
            }
            if (kt + PFD < nkt) {
                kload.load(kt + PFD, rk[f]);
                vload.load(kt + PFD, rv[f]);
            }
            __syncthreads();
            if ((causal && (kt + 1) * 64 > i0) || (kt + 1) * 64 > Nk) tile(kt, std::true_type{});
            else tile(kt, std::false_type{});
        }
    }
    __syncthreads();                                       // K / V images free: they become the output staging area
#pragma unroll
    for (int g = 0; g < QG; ++g) {
        float gs = gsum[g];
        gs += __shfl_xor(gs, 16, 64);
        gs += __shfl_xor(gs, 32, 64);
        // unmasked: rowsum(f) carries the constant N_k; the reference's constant is g0 (fastmax.py:271, fastmax_hack.py:21)
        const float gval = causal ? gs : gs - (float)Nk + prm.g0;
        if (qidx[g] < Nq && prm.g && q4 == 0) prm.g[(int64_t)bh * Nq + qidx[g]] = gval;
        const int row0 = i0 + 16 * (QG * w + g);
        store_tile16<DP>(smem + KI + w * (16 * DP * 4), oacc[g], 1.0f / gval, lane, prm.o, prm.out_dtype,
                         ((int64_t)bh * Nq + row0) * D, row0, Nq, D);
    }
}

template <int DP, int P, typename TIN, int NPP, int PFD>
static int launch_quad_pf(const QuadMfmaParams& prm, int B, hipStream_t stream) {
    constexpr int NP = InTraits<TIN>::NP, QG = quad_qg<DP, TIN, NPP>();
    constexpr int lds = (DP > 128 ? 2 : QG + 2) * NP * 64 * DP * 2;
    auto kern = fwd_quad_mfma_kernel<DP, P, TIN, NPP, PFD>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    dim3 grid((prm.Nq + 64 * QG - 1) / (64 * QG), B * prm.H), block(256);
    hipLaunchKernelGGL(kern, grid, block, lds, stream, prm);
    return (int)hipGetLastError();
}
// register prefetch depth: 1 tile.  Depth 2 / 3 measured 3-10 % slower (occupancy) in round 1.
template <int DP, int P, typename TIN, int NPP>
static int launch_quad_n(const QuadMfmaParams& prm, int B, hipStream_t stream) {
    return launch_quad_pf<DP, P, TIN, NPP, 1>(prm, B, stream);
}
template <int DP, int P, typename TIN>
static int launch_quad_t(const QuadMfmaParams& prm, int B, hipStream_t stream) {
    if constexpr (InTraits<TIN>::NP == 1) {
        if (prm.out_dtype != FASTMAX_F32) return launch_quad_n<DP, P, TIN, 1>(prm, B, stream);
    }
    return launch_quad_n<DP, P, TIN, 2>(prm, B, stream);
}
template <int P, typename TIN>
static int launch_quad_d(const QuadMfmaParams& prm, int B, hipStream_t stream) {
    if (prm.D > 128) return launch_quad_t<256, P, TIN>(prm, B, stream);
    return prm.D <= 64 ? launch_quad_t<64, P, TIN>(prm, B, stream) : launch_quad_t<128, P, TIN>(prm, B, stream);
}
template <typename TIN>
static int launch_quad_p(const QuadMfmaParams& prm, int B, int p, hipStream_t stream) {
    return p == 1 ? launch_quad_d<1, TIN>(prm, B, stream) : launch_quad_d<2, TIN>(prm, B, stream);
}

bool quad_mfma_supported(const fastmax_problem& p) {
    const int epl = p.in_dtype == FASTMAX_F32 ? 4 : 8;
    // a handful of queries against a short key range stays on the vector-ALU tiles; a single new token against a long KV
    // cache (generation, lit_gpt/model.py:464-466) must not: that kernel walks the keys with one query per wave
    // (measured 4.6 ms at N_k = 4096, 46 ms at 16 k; the matrix-core tile with 15 idle query rows takes 0.07 ms)
    return (p.D % epl) == 0 && p.D <= 256 && (p.Nq >= 16 || p.Nk >= 256);
}

int launch_fwd_quad_mfma(const FwdArgs& a) {
    if (!quad_mfma_supported(a.prob)) return FASTMAX_E_BAD_SHAPE;
    QuadMfmaParams prm{a.q, a.k, a.v, a.qs, a.ks, a.vs, a.o, a.g, a.prob.H, a.prob.Nq, a.prob.Nk, a.prob.D,
                       a.prob.causal, a.prob.out_dtype, a.prob.a, a.prob.g0};
    switch (a.prob.in_dtype) {
        case FASTMAX_F32: return launch_quad_p<float>(prm, a.prob.B, a.prob.p, a.stream);
        case FASTMAX_BF16: return launch_quad_p<bf16_t>(prm, a.prob.B, a.prob.p, a.stream);
        case FASTMAX_F16: return launch_quad_p<f16_t>(prm, a.prob.B, a.prob.p, a.stream);
    }
    return FASTMAX_E_BAD_DTYPE;
}

}  // namespace fastmax
