// fastmax backward (dQ, dK, dV; p = 1, 2; masked / unmasked) on 32x32x16 bf16 MFMA tiles (gfx950).
//
// Same dense gradients as fastmax_quad_mfma_bwd.hip (reference: attention_mechanisms/fastmax.py:383-691):
//   x_ij = a q_i.k_j,  P = f(x),  w_i = 1/g_i,  c_i = G_i.o_i,  u_ij = G_i.v_j
//   dS_ij = (u_ij - c_i) w_i f'(x_ij)        (only j <= i when causal)
//   dQ_i = a sum_j dS_ij k_j ;  dK_j = a sum_i dS_ij q_i ;  dV_j = sum_i P_ij w_i G_i
// in the structure of fastmax_quad32_mfma.hip: a wave owns 32 rows of the output (queries for dQ, keys for dK / dV),
// its own operand rows sit in registers as B fragments (loaded once from global memory), the other side streams
// through double-buffered LDS tiles with one barrier per tile, and the score-shaped tiles (S, U -> dS, P w) stay in
// registers as the B operand of the second product.  An LDS tile that is read by rows for one product and
// transposed for another (K for dQ; Q and G for dK / dV) uses the dual-use XOR image (img_off<DP, 3>).
// Scores are recomputed in both kernels, so there are no atomics and the result is bitwise reproducible.
//
// Round 3: the per-score vector work is folded into the matrix products.
//   * the prep pass hands over gt_i = w_i G_i (input dtype) and cw_i = c_i w_i, so (u_ij - c_i) w_i = gt_i.v_j - cw_i is
//     what the U chain delivers when its accumulator starts at -cw_i (a per-lane constant tile for dQ, four 16-byte LDS
//     reads per 32-query half for dK / dV), and dV_j = sum_i P_ij gt_i needs no row factor
//   * the S chain starts at u0 and delivers s = u0 + (scaled) q.k with f'(x) = e1 s and f(x) = e2 (s s + u0 u0) (p = 2)
//     or f = e1 s, f' = 1 (p = 1); the constants e1, e2 leave through the epilogue's per-row scale.  UNIT: the wave's own
//     rows (Q for dQ, K for dK / dV) are scaled by a on load and u0 is the inline constant 1.0, e1 = 1, e2 = 1/2;
//     otherwise u0 = 1/a in a register tile, e1 = a, e2 = a a / 2 (single-part operands whose a is not a power of two)
//   per score element: dQ one multiply (p = 2) or nothing (p = 1) + half a pack; dK / dV one fma + one multiply + one
//   pack (p = 2) or one pack (p = 1) -- against 3.5 and 7 before; no packed-f32 instructions (built without SLP).
//   * streamed tiles ride on buffer descriptors (BufTileLoader): no bounds compares, four integer adds per tile request
#include "fastmax_mfma32_common.h"

#include <cmath>
#include <cstdlib>
#include <type_traits>

namespace fastmax {

struct Quad32BwdParams {
    const void *q, *k, *v, *gt;     // gt = w G (B,H,Nq,D) contiguous, input dtype (prep pass)
    const float* cw;                // c_i w_i (B,H,Nq) from the prep pass
    Strides3 qs, ks, vs;
    void *dq, *dk, *dv;
    int H, BH, Nq, Nk, D, causal, grad_dtype, nblk;
    float a;
};

// workgroup id -> (head, block): whole heads per XCD (see fastmax_quad32_mfma.hip)
__device__ __forceinline__ void quad32_block(int L, int nblk, int BH, int& bh, int& blk) {
    if ((BH & 7) == 0) {
        const int x = L & 7, m = L >> 3;
        bh = x + 8 * (m / nblk);
        blk = m % nblk;
    } else {
        bh = L / nblk;
        blk = L % nblk;
    }
}

// f32x16 score-shaped tile -> two B fragments (16 k-rows each), NPP parts
template <int NPP> __device__ __forceinline__ void pack_tile(const float (&x)[16], Frag<NPP> (&f)[2]) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const f32x4 x0 = {x[8 * s], x[8 * s + 1], x[8 * s + 2], x[8 * s + 3]};
        const f32x4 x1 = {x[8 * s + 4], x[8 * s + 5], x[8 * s + 6], x[8 * s + 7]};
        if constexpr (NPP == 2) {
            bf16x4 h0, l0, h1, l1;
            split4(x0, h0, l0);
            split4(x1, h1, l1);
            f[s].p[0] = cat4(h0, h1);
            f[s].p[1] = cat4(l0, l1);
        } else {
            f[s].p[0] = cat4(to_bf16x4(x0), to_bf16x4(x1));
        }
    }
}

// first product of the S chain (accumulator starts at u0: inline 1.0 when UNIT, the register tile otherwise)
template <int NP, bool UNIT>
__device__ __forceinline__ f32x16 s_chain_head(const Frag<NP>& a, const Frag<NP>& b, const f32x16& cinit) {
    if constexpr (UNIT) return mfma32_parts_c1<NP, NP>(a, b);
    else return mfma32_parts<NP, NP>(a, b, cinit);
}

// ---- dQ: one wave = 32 queries, NW waves per workgroup, loop over 64-key tiles ---------------------------------------------
// MB = minimum workgroups per CU the register allocation must allow (2 -> 256 registers at NW = 4; 1 -> the whole file)
template <int DP, int P, typename TIN, int NW, int MB, bool UNIT>
__global__ __launch_bounds__(64 * NW, MB) void bwd32_dq_kernel(Quad32BwdParams prm) {
    constexpr int NP = InTraits<TIN>::NP, EPL = InTraits<TIN>::EPL, NPP = NP;
    constexpr int NT = 64 * NW, QT = 32 * NW;
    // K is read by rows (S chain) and transposed (dQ product), V by rows.  D <= 64: K in one dual-use XOR image.  D = 128: 48
    // fragment reads per tile would each need their own XOR-ed address (~100 integer instructions per tile), so K gets a
    // padded row image AND a padded transposed-read image: every fragment address is one lane constant + an immediate
    constexpr bool DUAL = DP == 128 && NP == 1;                            // two-part operands: the images would not fit
    constexpr int SWR = DUAL ? 1 : 3, SWT = DUAL ? 2 : 3;
    constexpr int IMG = img_bytes<DP, SWR>(), IMGT = DUAL ? img_bytes<DP, SWT>() : 0;
    constexpr int KTO = 2 * NP * IMG, STAGE = 2 * NP * IMG + NP * IMGT;     // K rows, V rows[, K transposed-read copy]
    constexpr int COLS = DP / EPL, RPP = NT / COLS, NPASS = 64 / RPP;
    static_assert(RPP <= 64 && NPASS >= 1, "staging map");
    constexpr int KS = DP / 16, DT = DP / 32;
    // the whole-file kernels at D = 128 keep the output accumulators in the AGPR half (mfma32_acc): the VGPRs are then free for
    // fragment reads in flight, without which every MFMA of a lone wave waits a full LDS round trip
    constexpr bool ACC_A = DP == 128 && MB == 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    int bh, qt;
    quad32_block(blockIdx.x, prm.nblk, prm.BH, bh, qt);
    const bool causal = prm.causal != 0;
    if (causal) qt = prm.nblk - 1 - qt;                                   // heaviest query blocks first
    const int b = bh / prm.H, hh = bh % prm.H;
    const int D = prm.D, Nq = prm.Nq, Nk = prm.Nk;
    const int i0 = qt * QT, qw0 = i0 + 32 * w, myq = qw0 + l31;
    const TIN* qb = reinterpret_cast<const TIN*>(prm.q) + (int64_t)b * prm.qs.sb + (int64_t)hh * prm.qs.sh;
    const TIN* kb = reinterpret_cast<const TIN*>(prm.k) + (int64_t)b * prm.ks.sb + (int64_t)hh * prm.ks.sh;
    const TIN* vb = reinterpret_cast<const TIN*>(prm.v) + (int64_t)b * prm.vs.sb + (int64_t)hh * prm.vs.sh;
    const TIN* gb = reinterpret_cast<const TIN*>(prm.gt) + (int64_t)bh * Nq * D;
    const int srow = tid / COLS, scol = tid % COLS;

    Frag<NP> qf[KS], gf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        if constexpr (P == 2) {
            if constexpr (UNIT) qf[ks] = load_q_frag_scaled<TIN>(qb, prm.qs.sn, myq, Nq, 16 * ks + 8 * h, D, prm.a);
            else qf[ks] = load_q_frag<TIN>(qb, prm.qs.sn, myq, Nq, 16 * ks + 8 * h, D);
        }
        gf[ks] = load_q_frag<TIN>(gb, D, myq, Nq, 16 * ks + 8 * h, D);
    }
    const int qc = myq < Nq ? myq : Nq - 1;
    const float ncw = -prm.cw[(int64_t)bh * Nq + qc];
    f32x16 ucinit, scinit;                                                // scinit is dead when UNIT or p = 1
#pragma unroll
    for (int i = 0; i < 16; ++i) { ucinit[i] = ncw; scinit[i] = 1.0f / prm.a; }
    const int klim = causal ? min(myq, Nk - 1) : Nk - 1;

    u32x4 rk[NPASS], rv[NPASS];
    const TileKernelLoader<TIN, NPASS, RPP> kload(kb, prm.ks.sn, Nk, D, srow, scol), vload(vb, prm.vs.sn, Nk, D, srow, scol);
    auto request = [&](int kt) __attribute__((always_inline)) {
        kload.load(kt, rk);
        vload.load(kt, rv);
    };
    auto commit = [&](int stage) __attribute__((always_inline)) {
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            stage_piece<DP, TIN, SWR>(smem, stage * STAGE, srow + ps * RPP, scol, rk[ps]);
            stage_piece<DP, TIN, SWR>(smem, stage * STAGE + NP * IMG, srow + ps * RPP, scol, rv[ps]);
            if constexpr (DUAL) stage_piece<DP, TIN, SWT>(smem, stage * STAGE + KTO, srow + ps * RPP, scol, rk[ps]);
        }
    };
    const int nkt = causal ? min((i0 + QT + 63) / 64, (Nk + 63) / 64) : (Nk + 63) / 64;

    f32x16 acc[DT];
#pragma unroll
    for (int mt = 0; mt < DT; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[mt][i] = 0.f;

    auto tile = [&](int kt, int stage, auto masked_tag) __attribute__((always_inline)) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        const int KI = stage * STAGE, VI = KI + NP * IMG, KT = DUAL ? KI + KTO : KI, k0 = kt * 64;
        constexpr int TIMG = DUAL ? IMGT : IMG;
        Frag<NPP> df[2][2];
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
            // all row fragments of this key half are requested before the two chains start: a read issued next to its MFMA
            // costs a full LDS round trip per matrix instruction when the wave is alone on its SIMD
            Frag<NP> kfr[KS], vfr[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    vfr[ks].p[p] = ld_row8<DP, SWR>(smem, VI + p * IMG, 32 * jt + l31, 2 * ks + h);
                    if constexpr (P == 2) kfr[ks].p[p] = ld_row8<DP, SWR>(smem, KI + p * IMG, 32 * jt + l31, 2 * ks + h);
                }
            f32x16 sc, u;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                // u[j][i] = v_j . gt_i - cw_i ;  s[j][i] = u0 + k_j . q_i
                u = mfma32_parts<NP, NP>(vfr[ks], gf[ks], ks == 0 ? ucinit : u);
                if constexpr (P == 2) sc = ks == 0 ? s_chain_head<NP, UNIT>(kfr[0], qf[0], scinit) : mfma32_parts<NP, NP>(kfr[ks], qf[ks], sc);
            }
            float ds[16];
            const int rel = klim - (k0 + 32 * jt) - 4 * h;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                float x = u[i];
                if constexpr (P == 2) x *= sc[i];
                if constexpr (MASKED) x = ((i & 3) + 8 * (i >> 2) <= rel) ? x : 0.f;
                ds[i] = x;
            }
            pack_tile<NPP>(ds, df[jt]);
        }
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
            Frag<NP> ktf[2][DT];
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int mt = 0; mt < DT; ++mt)
#pragma unroll
                    for (int p = 0; p < NP; ++p) ktf[s][mt].p[p] = ld_tr8_32<DP, SWT>(smem, KT + p * TIMG, 32 * jt + 16 * s, 32 * mt, lane);
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int mt = 0; mt < DT; ++mt) {                                    // dQ^T[m][i] += K[j][m] dS[j][i]
                    if constexpr (ACC_A) mfma32_parts_acc<NP, NPP>(acc[mt], ktf[s][mt], df[jt][s]);
                    else acc[mt] = mfma32_parts<NP, NPP>(ktf[s][mt], df[jt][s], acc[mt]);
                }
        }
    };
    auto advance = [&](int kt) __attribute__((always_inline)) {
        if (kt + 1 < nkt) {
            commit((kt & 1) ^ 1);
            if (kt + 2 < nkt) request(kt + 2);
        }
    };
    const int n_full = Nk / 64;
    const int n_plain = causal ? min((qw0 + 1) / 64, n_full) : n_full;
    const int n_act = causal ? min(nkt, (qw0 + 31) / 64 + 1) : nkt;

    request(0);
    commit(0);
    if (nkt > 1) request(1);
    __syncthreads();
    int kt = 0;
    for (; kt < n_plain; ++kt) {
        advance(kt);
        tile(kt, kt & 1, std::false_type{});
        __syncthreads();
    }
    for (; kt < n_act; ++kt) {
        advance(kt);
        tile(kt, kt & 1, std::true_type{});
        __syncthreads();
    }
    for (; kt < nkt; ++kt) {
        advance(kt);
        __syncthreads();
    }
    if constexpr (ACC_A) {
#pragma unroll
        for (int mt = 0; mt < DT; ++mt) acc_fence(acc[mt]);
    }
    // dQ_i = a e1 sum_j ds'_ij k_j  (e1 = a when the S chain carried unscaled scores)
    const float oscale = (P == 2 && !UNIT) ? prm.a * prm.a : prm.a;
    store_tile32_t<DT>(smem + w * 4096, acc, oscale, lane, prm.dq, prm.grad_dtype, (int64_t)bh * Nq, qw0, Nq, D);
}

// ---- dK, dV: one wave = 32 keys, NW waves per workgroup, loop over 64-query tiles -------------------------------------------
// D <= 64 bf16 fits two waves per SIMD; the other variants hold 128..256 registers of fragments and accumulators per wave
// and run one wave per SIMD with the accumulators in the AGPR half of the file
template <int DP, int P, typename TIN, int NW, bool UNIT>
__global__ __launch_bounds__(64 * NW, (DP == 64 && InTraits<TIN>::NP == 1 && NW == 4) ? 2 : 1) void bwd32_dkv_kernel(Quad32BwdParams prm) {
    constexpr int NP = InTraits<TIN>::NP, EPL = InTraits<TIN>::EPL, NPP = NP;
    constexpr int NT = 64 * NW, KT = 32 * NW;
    // Q and gt are both read by rows (S, U chains) and transposed (dK, dV products): one dual-use XOR image each at D <= 64,
    // a padded row image + a padded transposed-read image each at D = 128 (see bwd32_dq_kernel)
    constexpr bool DUAL = DP == 128 && NP == 1;        // two-part operands: the images would not fit
    constexpr int SWR = DUAL ? 1 : 3, SWT = DUAL ? 2 : 3;
    constexpr int IMG = img_bytes<DP, SWR>(), IMGT = DUAL ? img_bytes<DP, SWT>() : 0;
    constexpr int TRO = 2 * NP * IMG, CWO = TRO + 2 * NP * IMGT, STAGE = CWO + 256;   // Q rows, gt rows[, Q tr, gt tr], -cw[64]
    constexpr int COLS = DP / EPL, RPP = NT / COLS, NPASS = 64 / RPP;
    static_assert(RPP <= 64 && NPASS >= 1, "staging map");
    constexpr int KS = DP / 16, DT = DP / 32;
    constexpr bool ACC_A = DP == 128 && NW == 4;                          // accumulators in the AGPR half (see bwd32_dq_kernel)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    int bh, kb_;
    quad32_block(blockIdx.x, prm.nblk, prm.BH, bh, kb_);                    // causal: low key blocks are the heavy ones, first
    const bool causal = prm.causal != 0;
    const int b = bh / prm.H, hh = bh % prm.H;
    const int D = prm.D, Nq = prm.Nq, Nk = prm.Nk;
    const int j0 = kb_ * KT, jw0 = j0 + 32 * w, myk = jw0 + l31;
    const TIN* qb = reinterpret_cast<const TIN*>(prm.q) + (int64_t)b * prm.qs.sb + (int64_t)hh * prm.qs.sh;
    const TIN* kb = reinterpret_cast<const TIN*>(prm.k) + (int64_t)b * prm.ks.sb + (int64_t)hh * prm.ks.sh;
    const TIN* vb = reinterpret_cast<const TIN*>(prm.v) + (int64_t)b * prm.vs.sb + (int64_t)hh * prm.vs.sh;
    const TIN* gb = reinterpret_cast<const TIN*>(prm.gt) + (int64_t)bh * Nq * D;
    const int srow = tid / COLS, scol = tid % COLS;

    Frag<NP> kf[KS], vf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        if constexpr (UNIT) kf[ks] = load_q_frag_scaled<TIN>(kb, prm.ks.sn, myk, Nk, 16 * ks + 8 * h, D, prm.a);
        else kf[ks] = load_q_frag<TIN>(kb, prm.ks.sn, myk, Nk, 16 * ks + 8 * h, D);
        vf[ks] = load_q_frag<TIN>(vb, prm.vs.sn, myk, Nk, 16 * ks + 8 * h, D);
    }
    const float u0 = UNIT ? 1.0f : 1.0f / prm.a, c0 = u0 * u0;
    f32x16 scinit;                                                        // dead when UNIT
#pragma unroll
    for (int i = 0; i < 16; ++i) scinit[i] = u0;
    // rows (queries) of a score tile this lane's key may see: qlo <= query <= Nq - 1 (nothing for a key past N_k)
    const int qlo = myk < Nk ? (causal ? myk : 0) : 0x7fffffff;

    u32x4 rq[NPASS], rg[NPASS];
    float rcw = 0.f;
    const TileKernelLoader<TIN, NPASS, RPP> qload(qb, prm.qs.sn, Nq, D, srow, scol), gload(gb, D, Nq, D, srow, scol);
    auto request = [&](int it) __attribute__((always_inline)) {
        qload.load(it, rq);
        gload.load(it, rg);
        if (tid < 64) {
            const int gi = it * 64 + tid, gc = gi < Nq ? gi : Nq - 1;
            rcw = prm.cw[(int64_t)bh * Nq + gc];         // raw: negated in commit() -- any arithmetic here needs the value at once,
                                                          // i.e. an s_waitcnt vmcnt(0) right behind the tile requests
        }
    };
    auto commit = [&](int stage) __attribute__((always_inline)) {
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            stage_piece<DP, TIN, SWR>(smem, stage * STAGE, srow + ps * RPP, scol, rq[ps]);
            stage_piece<DP, TIN, SWR>(smem, stage * STAGE + NP * IMG, srow + ps * RPP, scol, rg[ps]);
            if constexpr (DUAL) {
                stage_piece<DP, TIN, SWT>(smem, stage * STAGE + TRO, srow + ps * RPP, scol, rq[ps]);
                stage_piece<DP, TIN, SWT>(smem, stage * STAGE + TRO + NP * IMGT, srow + ps * RPP, scol, rg[ps]);
            }
        }
        if (tid < 64) reinterpret_cast<float*>(smem + stage * STAGE + CWO)[tid] = -rcw;
    };
    const int nqt = (Nq + 63) / 64;
    const int it0 = causal ? min(j0 / 64, nqt) : 0;

    f32x16 dkacc[DT], dvacc[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) { dkacc[t][i] = 0.f; dvacc[t][i] = 0.f; }

    auto tile_plain = [&](int it, int stage, auto masked_tag) __attribute__((always_inline)) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        const int QI = stage * STAGE, GI = QI + NP * IMG, CS = QI + CWO;
        const int QTI = DUAL ? QI + TRO : QI, GTI = DUAL ? QTI + NP * IMGT : GI;
        constexpr int TIMG = DUAL ? IMGT : IMG;
#pragma unroll
        for (int qs = 0; qs < 2; ++qs) {
            // -cw of this half's 32 queries in accumulator order: row(i) = (i&3) + 8(i>>2) + 4h
            f32x16 ucinit;
#pragma unroll
            for (int ig = 0; ig < 4; ++ig) {
                const f32x4 c4 = *reinterpret_cast<const f32x4*>(smem + CS + (32 * qs + 8 * ig + 4 * h) * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) ucinit[4 * ig + e] = c4[e];
            }
            Frag<NP> qrf[KS], grf[KS];                                  // requested up front (see bwd32_dq_kernel)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    qrf[ks].p[p] = ld_row8<DP, SWR>(smem, QI + p * IMG, 32 * qs + l31, 2 * ks + h);
                    grf[ks].p[p] = ld_row8<DP, SWR>(smem, GI + p * IMG, 32 * qs + l31, 2 * ks + h);
                }
            f32x16 sc, u;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                // s[i][j] = u0 + q_i . k_j (rows = queries, column = this lane's key);  u[i][j] = gt_i . v_j - cw_i
                sc = ks == 0 ? s_chain_head<NP, UNIT>(qrf[0], kf[0], scinit) : mfma32_parts<NP, NP>(qrf[ks], kf[ks], sc);
                u = mfma32_parts<NP, NP>(grf[ks], vf[ks], ks == 0 ? ucinit : u);
            }
            Frag<NPP> pwf[2], dsf[2];
            {
                float pw[16], ds[16];
                const int lo = qlo - (it * 64 + 32 * qs) - 4 * h, hi = Nq - 1 - (it * 64 + 32 * qs) - 4 * h;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    float pwv = sc[i], dsv = u[i];
                    if constexpr (P == 2) {
                        pwv = fmaf(sc[i], sc[i], c0);
                        dsv *= sc[i];
                    }
                    if constexpr (MASKED) {
                        const int r = (i & 3) + 8 * (i >> 2);
                        const bool keep = r >= lo && r <= hi;
                        pwv = keep ? pwv : 0.f;
                        dsv = keep ? dsv : 0.f;
                    }
                    pw[i] = pwv;
                    ds[i] = dsv;
                }
                pack_tile<NPP>(pw, pwf);
                pack_tile<NPP>(ds, dsf);
            }
            Frag<NP> gtf[2][DT], qtf[2][DT];
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int t = 0; t < DT; ++t)
#pragma unroll
                    for (int p = 0; p < NP; ++p) {
                        gtf[s][t].p[p] = ld_tr8_32<DP, SWT>(smem, GTI + p * TIMG, 32 * qs + 16 * s, 32 * t, lane);
                        qtf[s][t].p[p] = ld_tr8_32<DP, SWT>(smem, QTI + p * TIMG, 32 * qs + 16 * s, 32 * t, lane);
                    }
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int t = 0; t < DT; ++t) {
                    // dV^T[d][j] += gt[i][d] P'_ij ;  dK^T[m][j] += Q[i][m] dS'_ij
                    if constexpr (ACC_A) {
                        mfma32_parts_acc<NP, NPP>(dvacc[t], gtf[s][t], pwf[s]);
                        mfma32_parts_acc<NP, NPP>(dkacc[t], qtf[s][t], dsf[s]);
                    } else {
                        dvacc[t] = mfma32_parts<NP, NPP>(gtf[s][t], pwf[s], dvacc[t]);
                        dkacc[t] = mfma32_parts<NP, NPP>(qtf[s][t], dsf[s], dkacc[t]);
                    }
                }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // Single-part operands: the same tile as a software pipeline over its two 32-query halves, so that the vector ALU's turn on
    // one half's score tiles runs beside the other half's matrix instructions (a wave alone on its SIMD has nobody else to fill
    // the matrix pipe while it multiplies and packs):
    //     S,U(h0)  |  S,U(h1) || f(h0)  |  dV,dK += (h0) || f(h1)  |  dV,dK += (h1)
    // In the two middle phases one score element is processed after every matrix instruction (sched_barrier fences pin the
    // order); the fragments of a phase are requested one phase ahead.
    auto tile_pipe = [&](int it, int stage, auto masked_tag) __attribute__((always_inline)) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        const int QI = stage * STAGE, GI = QI + NP * IMG, CS = QI + CWO;
        const int QTI = DUAL ? QI + TRO : QI, GTI = DUAL ? QTI + NP * IMGT : GI;
        constexpr int TIMG = DUAL ? IMGT : IMG;
        constexpr int NCH = 2 * KS, NAC = 4 * DT;                   // matrix instructions of a chain phase / an accumulate phase
        Frag<NP> qrf[2][KS], grf[2][KS], gtf[2][2][DT], qtf[2][2][DT];
        f32x16 sc[2], u[2];
        float pw[2][16], ds[2][16];
        Frag<NPP> pwf[2][2], dsf[2][2];
        auto readG = [&](int qs) __attribute__((always_inline)) {
#pragma unroll
            for (int ig = 0; ig < 4; ++ig) {
                const f32x4 c4 = *reinterpret_cast<const f32x4*>(smem + CS + (32 * qs + 8 * ig + 4 * h) * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) u[qs][4 * ig + e] = c4[e];
            }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                qrf[qs][ks].p[0] = ld_row8<DP, SWR>(smem, QI, 32 * qs + l31, 2 * ks + h);
                grf[qs][ks].p[0] = ld_row8<DP, SWR>(smem, GI, 32 * qs + l31, 2 * ks + h);
            }
        };
        auto readT = [&](int qs, int s_) __attribute__((always_inline)) {
#pragma unroll
            for (int t = 0; t < DT; ++t) {
                gtf[qs][s_][t].p[0] = ld_tr8_32<DP, SWT>(smem, GTI, 32 * qs + 16 * s_, 32 * t, lane);
                qtf[qs][s_][t].p[0] = ld_tr8_32<DP, SWT>(smem, QTI, 32 * qs + 16 * s_, 32 * t, lane);
            }
        };
        // m-th matrix instruction of a chain phase: k-step m/2 of S (even m) or U (odd m)
        auto chain_step = [&](int qs, int m) __attribute__((always_inline)) {
            const int ks = m >> 1;
            if ((m & 1) == 0) sc[qs] = ks == 0 ? s_chain_head<NP, UNIT>(qrf[qs][0], kf[0], scinit) : mfma32_parts<NP, NP>(qrf[qs][ks], kf[ks], sc[qs]);
            else u[qs] = mfma32_parts<NP, NP>(grf[qs][ks], vf[ks], u[qs]);
        };
        // m-th matrix instruction of an accumulate phase: (s, t, dV | dK)
        auto acc_step = [&](int qs, int m) __attribute__((always_inline)) {
            const int s_ = m / (2 * DT), t = (m >> 1) % DT;
            if ((m & 1) == 0) {
                if constexpr (ACC_A) mfma32_parts_acc<NP, NPP>(dvacc[t], gtf[qs][s_][t], pwf[qs][s_]);
                else dvacc[t] = mfma32_parts<NP, NPP>(gtf[qs][s_][t], pwf[qs][s_], dvacc[t]);
            } else {
                if constexpr (ACC_A) mfma32_parts_acc<NP, NPP>(dkacc[t], qtf[qs][s_][t], dsf[qs][s_]);
                else dkacc[t] = mfma32_parts<NP, NPP>(qtf[qs][s_][t], dsf[qs][s_], dkacc[t]);
            }
        };
        auto valu_elem = [&](int qs, int i) __attribute__((always_inline)) {
            float pwv = sc[qs][i], dsv = u[qs][i];
            if constexpr (P == 2) {
                pwv = fmaf(sc[qs][i], sc[qs][i], c0);
                dsv *= sc[qs][i];
            }
            if constexpr (MASKED) {
                const int lo = qlo - (it * 64 + 32 * qs) - 4 * h, hi = Nq - 1 - (it * 64 + 32 * qs) - 4 * h;
                const int r = (i & 3) + 8 * (i >> 2);
                const bool keep = r >= lo && r <= hi;
                pwv = keep ? pwv : 0.f;
                dsv = keep ? dsv : 0.f;
            }
            pw[qs][i] = pwv;
            ds[qs][i] = dsv;
        };
        auto pack_half = [&](int qs, int s_) __attribute__((always_inline)) {
            const f32x4 p0 = {pw[qs][8 * s_], pw[qs][8 * s_ + 1], pw[qs][8 * s_ + 2], pw[qs][8 * s_ + 3]};
            const f32x4 p1 = {pw[qs][8 * s_ + 4], pw[qs][8 * s_ + 5], pw[qs][8 * s_ + 6], pw[qs][8 * s_ + 7]};
            const f32x4 d0 = {ds[qs][8 * s_], ds[qs][8 * s_ + 1], ds[qs][8 * s_ + 2], ds[qs][8 * s_ + 3]};
            const f32x4 d1 = {ds[qs][8 * s_ + 4], ds[qs][8 * s_ + 5], ds[qs][8 * s_ + 6], ds[qs][8 * s_ + 7]};
            pwf[qs][s_].p[0] = cat4(to_bf16x4(p0), to_bf16x4(p1));
            dsf[qs][s_].p[0] = cat4(to_bf16x4(d0), to_bf16x4(d1));
        };
        // the 16 score elements of a half spread over the NM (8 or 16) matrix instructions of the phase beside it
        auto valu_slice = [&](int qs, int m, auto nm_tag) __attribute__((always_inline)) {
            constexpr int EPM = 16 / decltype(nm_tag)::value;
            static_assert(EPM * decltype(nm_tag)::value == 16, "phase length");
#pragma unroll
            for (int e = 0; e < EPM; ++e) {
                const int i = m * EPM + e;
                valu_elem(qs, i);
                if (i == 7) pack_half(qs, 0);
                if (i == 15) pack_half(qs, 1);
            }
        };
#define Q32_FENCE() __builtin_amdgcn_sched_barrier(0)
        // reads are requested half a phase ahead of their first use: early enough for a lone wave, late enough that two full
        // fragment sets are never alive together (D = 128: 64 registers each)
        readG(0);
        Q32_FENCE();
#pragma unroll
        for (int m = 0; m < NCH; ++m) {
            if (m == NCH / 2) readG(1);
            chain_step(0, m);
            if (m == NCH / 2 - 1) Q32_FENCE();
        }
        Q32_FENCE();
#pragma unroll
        for (int m = 0; m < NCH; ++m) {
            if (m == NCH / 2) readT(0, 0);
            chain_step(1, m);
            valu_slice(0, m, std::integral_constant<int, NCH>{});
            Q32_FENCE();
        }
#pragma unroll
        for (int m = 0; m < NAC; ++m) {
            if (m == 0) readT(0, 1);
            if (m == NAC / 2) readT(1, 0);
            acc_step(0, m);
            valu_slice(1, m, std::integral_constant<int, NAC>{});
            Q32_FENCE();
        }
#pragma unroll
        for (int m = 0; m < NAC; ++m) {
            if (m == 0) readT(1, 1);
            acc_step(1, m);
        }
#undef Q32_FENCE
    };
    auto tile = [&](int it, int stage, auto masked_tag) __attribute__((always_inline)) {
        // one wave per SIMD (D = 128, single-part operands): the explicit pipeline; two waves per SIMD interleave by themselves
        // and lose 2 % to the fences
        if constexpr (NP == 1 && DP == 128) tile_pipe(it, stage, masked_tag);
        else tile_plain(it, stage, masked_tag);
    };
    auto advance = [&](int it) __attribute__((always_inline)) {
        if (it + 1 < nqt) {
            commit(((it - it0) & 1) ^ 1);
            if (it + 2 < nqt) request(it + 2);
        }
    };
    // per wave: query tiles [it0, it_a) lie wholly before this wave's keys (causal), [it_a, it_p) touch the diagonal,
    // [it_p, n_fullq) are plain, the last one is partial when N_q % 64 != 0; a key block that runs past N_k masks all
    const int n_fullq = Nq / 64;
    int it_a = it0, it_p = it0;
    if (causal) {
        it_a = min(max(it0, jw0 / 64), nqt);
        it_p = min(max(it_a, (jw0 + 31 + 63) / 64), nqt);
    }
    if (j0 + KT > Nk) it_p = nqt;
    const int it_q = max(it_p, min(n_fullq, nqt));

    if (it0 < nqt) {
        request(it0);
        commit(0);
        if (it0 + 1 < nqt) request(it0 + 1);
    }
    __syncthreads();
    int it = it0;
    for (; it < it_a; ++it) {
        advance(it);
        __syncthreads();
    }
    for (; it < it_p; ++it) {
        advance(it);
        tile(it, (it - it0) & 1, std::true_type{});
        __syncthreads();
    }
    for (; it < it_q; ++it) {
        advance(it);
        tile(it, (it - it0) & 1, std::false_type{});
        __syncthreads();
    }
    for (; it < nqt; ++it) {
        advance(it);
        tile(it, (it - it0) & 1, std::true_type{});
        __syncthreads();
    }
    if constexpr (ACC_A) {
#pragma unroll
        for (int t = 0; t < DT; ++t) { acc_fence(dkacc[t]); acc_fence(dvacc[t]); }
    }
    // dK_j = a e1 sum_i dS'_ij q_i ;  dV_j = e2 sum_i P'_ij gt_i   (e1, e2: see the file header)
    const float e1 = (P == 2 && !UNIT) ? prm.a : 1.0f;
    const float e2 = (P == 1 ? 1.0f : 0.5f) * (UNIT ? 1.0f : (P == 1 ? prm.a : prm.a * prm.a));
    store_tile32_t<DT>(smem + w * 4096, dkacc, prm.a * e1, lane, prm.dk, prm.grad_dtype, (int64_t)bh * Nk, jw0, Nk, D);
    store_tile32_t<DT>(smem + w * 4096, dvacc, e2, lane, prm.dv, prm.grad_dtype, (int64_t)bh * Nk, jw0, Nk, D);
}

template <int DP, int P, typename TIN, int NW, int MB, bool UNIT>
static int launch_bwd32_dq(Quad32BwdParams prm, hipStream_t stream) {
    constexpr int NP = InTraits<TIN>::NP;
    constexpr int st_q = 2 * ((DP == 128 && NP == 1) ? NP * (2 * img_bytes<DP, 1>() + img_bytes<DP, 2>()) : 2 * NP * img_bytes<DP, 3>()), epi = NW * 4096;
    constexpr int lds_q = st_q > epi ? st_q : epi;
    auto kq = bwd32_dq_kernel<DP, P, TIN, NW, MB, UNIT>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kq), hipFuncAttributeMaxDynamicSharedMemorySize, lds_q);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    prm.nblk = (prm.Nq + 32 * NW - 1) / (32 * NW);
    hipLaunchKernelGGL(kq, dim3(prm.nblk * prm.BH), dim3(64 * NW), lds_q, stream, prm);
    return (int)hipGetLastError();
}
template <int DP, int P, typename TIN, int NW, bool UNIT>
static int launch_bwd32_dkv(Quad32BwdParams prm, hipStream_t stream) {
    constexpr int NP = InTraits<TIN>::NP;
    constexpr int st_kv = 2 * (((DP == 128 && NP == 1) ? 2 * NP * (img_bytes<DP, 1>() + img_bytes<DP, 2>()) : 2 * NP * img_bytes<DP, 3>()) + 256), epi = NW * 4096;
    constexpr int lds_kv = st_kv > epi ? st_kv : epi;
    auto kkv = bwd32_dkv_kernel<DP, P, TIN, NW, UNIT>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kkv), hipFuncAttributeMaxDynamicSharedMemorySize, lds_kv);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    prm.nblk = (prm.Nk + 32 * NW - 1) / (32 * NW);
    hipLaunchKernelGGL(kkv, dim3(prm.nblk * prm.BH), dim3(64 * NW), lds_kv, stream, prm);
    return (int)hipGetLastError();
}
template <int DP, int P, typename TIN, bool UNIT>
static int launch_bwd32_n(const Quad32BwdParams& prm, hipStream_t stream) {
    constexpr int NP = InTraits<TIN>::NP;
    // measured choices (profiles/r01_quad32_p2_bf16.md, r03_p2_tiles.md): two-part operands at D <= 64 take eight waves per
    // workgroup; bf16 at D <= 64 four waves and two workgroups per CU (256 registers); D = 128 four waves with the whole
    // register file (the 256-register cap spills the accumulators into the tile loop)
    int rc;
    if constexpr (DP == 64 && NP == 2) rc = launch_bwd32_dq<DP, P, TIN, 8, 1, UNIT>(prm, stream);
    else if constexpr (DP == 64) rc = launch_bwd32_dq<DP, P, TIN, 4, 2, UNIT>(prm, stream);
    else rc = launch_bwd32_dq<DP, P, TIN, 4, 1, UNIT>(prm, stream);
    if (rc) return rc;
    if constexpr (DP == 64 && NP == 2) return launch_bwd32_dkv<DP, P, TIN, 8, UNIT>(prm, stream);
    else return launch_bwd32_dkv<DP, P, TIN, 4, UNIT>(prm, stream);
}
static bool is_pow2(float a) { int e; return a > 0.f && frexpf(a, &e) == 0.5f; }
template <int DP, int P, typename TIN>
static int launch_bwd32_u(const Quad32BwdParams& prm, hipStream_t stream) {
    // split (fp32 / fp16) operands are scaled in fp32 before the split; bf16 rows only by a power of two (exact)
    if constexpr (InTraits<TIN>::NP == 2) return launch_bwd32_n<DP, P, TIN, true>(prm, stream);
    else return is_pow2(prm.a) ? launch_bwd32_n<DP, P, TIN, true>(prm, stream) : launch_bwd32_n<DP, P, TIN, false>(prm, stream);
}
template <int P, typename TIN>
static int launch_bwd32_d(const Quad32BwdParams& prm, hipStream_t stream) {
    return prm.D <= 64 ? launch_bwd32_u<64, P, TIN>(prm, stream) : launch_bwd32_u<128, P, TIN>(prm, stream);
}
template <typename TIN>
static int launch_bwd32_p(const Quad32BwdParams& prm, int p, hipStream_t stream) {
    return p == 1 ? launch_bwd32_d<1, TIN>(prm, stream) : launch_bwd32_d<2, TIN>(prm, stream);
}

bool quad32_bwd_supported(const fastmax_problem& p) {
    static const int mode = [] { const char* e = getenv("FASTMAX_QUAD32_BWD"); return e ? atoi(e) : 1; }();
    if (!mode) return false;
    const int epl = p.in_dtype == FASTMAX_F32 ? 4 : 8;
    return (p.D % epl) == 0 && p.D <= 128 && p.Nq >= 256 && p.Nk >= 256 && (int64_t)p.B * p.H * ((max(p.Nq, p.Nk) + 127) / 128) <= 0x7fffffff;
}
// every streamed (b,h) slab fits the 31-bit buffer offsets of BufTileLoader
bool quad32_bwd_layout_ok(const BwdArgs& a) {
    const int eb = a.prob.in_dtype == FASTMAX_F32 ? 4 : 2;
    return quad32_span_ok(a.qs.sn, a.prob.Nq, a.prob.D, eb) && quad32_span_ok(a.ks.sn, a.prob.Nk, a.prob.D, eb) &&
           quad32_span_ok(a.vs.sn, a.prob.Nk, a.prob.D, eb) && quad32_span_ok(a.prob.D, a.prob.Nq, a.prob.D, eb);
}

// the prep pass (bwd_prep_kernel of fastmax_quad_mfma_bwd.hip, gt mode) has left cw (B,H,Nq) at the head of a.workspace
// and gt = w G (B,H,Nq,D) at quad32_bwd_gt_offset
int launch_bwd_quad32_main(const BwdArgs& a) {
    Quad32BwdParams prm{a.q, a.k, a.v, reinterpret_cast<const char*>(a.workspace) + quad32_bwd_gt_offset(a.prob),
                        reinterpret_cast<const float*>(a.workspace), a.qs, a.ks, a.vs,
                        a.dq, a.dk, a.dv, a.prob.H, a.prob.B * a.prob.H, a.prob.Nq, a.prob.Nk, a.prob.D, a.prob.causal,
                        a.prob.in_dtype, 0, a.prob.a};
    switch (a.prob.in_dtype) {
        case FASTMAX_F32: return launch_bwd32_p<float>(prm, a.prob.p, a.stream);
        case FASTMAX_BF16: return launch_bwd32_p<bf16_t>(prm, a.prob.p, a.stream);
        case FASTMAX_F16: return launch_bwd32_p<f16_t>(prm, a.prob.p, a.stream);
    }
    return FASTMAX_E_BAD_DTYPE;
}

}  // namespace fastmax
