// fastmax forward, quadratic evaluation, on 32x32x16 bf16 MFMA tiles (gfx950).
//
// Same function as fastmax_quad_mfma.hip (o_i = sum_j f(a q_i.k_j) v_j / g_i; reference: attention_mechanisms/fastmax.py:184-322,
// both p, masked / unmasked, N_q != N_k); this is the throughput form for long sequences:
//   * a wave owns QB blocks of 32 queries and NW waves (4 or 8) share every 64-key K / V tile, so one LDS fragment read
//     (1 KiB) feeds QB 32x32x16 MFMAs (32 K MAC each) instead of a 16x16x32 one (16 K MAC)
//   * Q fragments are loaded once from global memory straight into registers (no Q image)
//   * K / V tiles are double-buffered in LDS; tile t+1 is written from registers right after the single barrier of
//     tile t and tile t+2 is requested immediately (one barrier per tile, loads one full tile ahead)
//   * S^T = K Q^T keeps the key on the accumulator row, so P = f(a S) in registers is already the B operand of
//     O^T += V^T P^T; V^T fragments come from ds_read_b64_tr_b16 with the matching key order
//   * f costs ONE vector instruction per element (round 3).  With u = 1/a + q.k the polynomial is a perfect-square form:
//         p = 1:  1 + a s         = a u
//         p = 2:  1 + a s + (a s)^2 / 2 = (a^2 / 2) (u^2 + 1/a^2)
//     u comes straight out of the S^T chain, whose accumulator starts at the constant 1/a instead of 0 (when a is a
//     power of two, or the operands are split fp32 / fp16 values, Q is scaled by a on load and the constant is the inline
//     1.0); the common factor (a or a^2/2) cancels in o = F / g and is applied to the stored g only.  Per score element
//     the tile loop issues one v_fma (u u + c), one v_add (row sum) and half a v_cvt_pk: 2.5 vector instructions against
//     5.2 before, none of them packed-f32 (v_pk_* beside MFMAs costs 3-4x its issue slot on gfx950).
//   * K / V tiles ride on buffer descriptors whose record count is the head's own byte range: rows past N_k and padded
//     head columns read as zero in hardware, a tile request is four loads and four integer adds
//   * QB = 2 (bf16, D <= 64): two independent query blocks per wave give the wave its own matrix work to issue while
//     the vector ALU turns the other block's scores into P, and halve fragment reads, staging and barriers per MFMA
// 32x32x16 layouts (A: row = lane&31, k = 8(lane>>5)+j; B: col = lane&31, same k; C: col = lane&31,
// row(i) = (i&3) + 8(i>>2) + 4(lane>>5)).
#include "fastmax_mfma32_common.h"

#include <cmath>
#include <cstdlib>
#include <type_traits>

namespace fastmax {

struct Quad32Params {
    const void *q, *k, *v;
    Strides3 qs, ks, vs;
    void* o;
    float* g;
    int H, BH, Nq, Nk, D, causal, out_dtype, nqt;
    float a, g0;
};

template <int DP, typename TIN, int NPP, int NW, int QB> constexpr int quad32_min_blocks() {
    if (QB == 2) return 2;                                        // ~230 registers: two 4-wave workgroups per CU
    // bf16 D<=64 with a 16-bit result needs ~150 registers: three 4-wave workgroups per CU; everything else two waves / SIMD
    if (DP == 64 && InTraits<TIN>::NP == 1 && NPP == 1) return NW == 8 ? 1 : 3;
    return NW == 8 ? 1 : 2;
}

// 1-D grid of nqt * BH workgroups; block = 64 NW threads; dynamic LDS = max(2 stages of K,V images, NW * 4 KiB)
// UNIT: Q is scaled by a on load and the score chain starts at the inline constant 1.0 (u = 1 + a q.k); otherwise the
// chain starts at a register tile holding 1/a (u = 1/a + q.k) -- single-part operands whose scale is not a power of two
// SCHED: issue order of a tile, see tile1 / tile2
template <int DP, int P, typename TIN, int NPP, int NW, bool UNIT, int SCHED, int QB>
__global__ __launch_bounds__(64 * NW, (quad32_min_blocks<DP, TIN, NPP, NW, QB>())) void fwd_quad32_kernel(Quad32Params prm) {
    constexpr int NP = InTraits<TIN>::NP, EPL = InTraits<TIN>::EPL;
    constexpr int NT = 64 * NW, QW = 32 * QB, QT = QW * NW;
    constexpr int KIMG = img_bytes<DP, 1>(), VIMG = img_bytes<DP, 2>(), STAGE = NP * (KIMG + VIMG);
    constexpr int COLS = DP / EPL, RPP = NT / COLS, NPASS = 64 / RPP;
    static_assert(RPP <= 64 && NPASS >= 1, "staging map");
    static_assert(QB == 1 || (NP == 1 && NPP == 1), "two query blocks per wave: single-part operands only");
    constexpr int KS = DP / 16, DT = DP / 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    // workgroup -> (head, query block): consecutive workgroup ids go round-robin over the 8 XCDs, so give every XCD
    // whole heads (their K / V then stay in that XCD's L2); causal: heaviest query blocks of a head first
    int bh, qt;
    {
        const int L = blockIdx.x, nqt = prm.nqt;
        if ((prm.BH & 7) == 0) {
            const int x = L & 7, m = L >> 3;
            bh = x + 8 * (m / nqt);
            qt = m % nqt;
        } else {
            bh = L / nqt;
            qt = L % nqt;
        }
        if (prm.causal) qt = nqt - 1 - qt;
    }
    const int b = bh / prm.H, hh = bh % prm.H;
    const int D = prm.D, Nq = prm.Nq, Nk = prm.Nk;
    const bool causal = prm.causal != 0;
    const int i0 = qt * QT, qw0 = i0 + QW * w;
    const TIN* qb = reinterpret_cast<const TIN*>(prm.q) + (int64_t)b * prm.qs.sb + (int64_t)hh * prm.qs.sh;
    const TIN* kb = reinterpret_cast<const TIN*>(prm.k) + (int64_t)b * prm.ks.sb + (int64_t)hh * prm.ks.sh;
    const TIN* vb = reinterpret_cast<const TIN*>(prm.v) + (int64_t)b * prm.vs.sb + (int64_t)hh * prm.vs.sh;
    const int srow = tid / COLS, scol = tid % COLS;

    Frag<NP> qf[QB][KS];
    int klim[QB];                                                     // last key a lane's query may see
#pragma unroll
    for (int qb_ = 0; qb_ < QB; ++qb_) {
        const int myq = qw0 + 32 * qb_ + l31;
        klim[qb_] = causal ? min(myq, Nk - 1) : Nk - 1;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if constexpr (UNIT) qf[qb_][ks] = load_q_frag_scaled<TIN>(qb, prm.qs.sn, myq, Nq, 16 * ks + 8 * h, D, prm.a);
            else qf[qb_][ks] = load_q_frag<TIN>(qb, prm.qs.sn, myq, Nq, 16 * ks + 8 * h, D);
        }
    }

    u32x4 rk[NPASS], rv[NPASS];
    const TileKernelLoader<TIN, NPASS, RPP> kload(kb, prm.ks.sn, Nk, D, srow, scol), vload(vb, prm.vs.sn, Nk, D, srow, scol);
    auto request = [&](int kt) __attribute__((always_inline)) {
        kload.load(kt, rk);
        vload.load(kt, rv);
    };
    auto commit = [&](int stage) __attribute__((always_inline)) {
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            stage_piece<DP, TIN, 1>(smem, stage * STAGE, srow + ps * RPP, scol, rk[ps]);
            stage_piece<DP, TIN, 2>(smem, stage * STAGE + NP * KIMG, srow + ps * RPP, scol, rv[ps]);
        }
    };
    const int nkt = causal ? min((i0 + QT + 63) / 64, (Nk + 63) / 64) : (Nk + 63) / 64;

    f32x16 oacc[QB][DT];
    float gsum[QB][4];
#pragma unroll
    for (int qb_ = 0; qb_ < QB; ++qb_) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int i = 0; i < 16; ++i) oacc[qb_][dt][i] = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) gsum[qb_][i] = 0.f;
    }
    // u = u0 + (scaled) q.k out of the score chain; f = fscale * (p == 1 ? u : u u + c0)
    const float u0 = UNIT ? 1.0f : 1.0f / prm.a, c0 = u0 * u0;
    const float fscale = (P == 1 ? 1.0f : 0.5f) * (UNIT ? 1.0f : (P == 1 ? prm.a : prm.a * prm.a));
    f32x16 cinit;                                                     // dead when UNIT
#pragma unroll
    for (int i = 0; i < 16; ++i) cinit[i] = u0;

    // f of one S^T tile (32 keys x 32 queries of block qb_) -> the two B fragments (16 keys each) of the P^T operand
    auto poly = [&](const f32x16& sc, int qb_, int key0, auto masked_tag, Frag<NPP> (&pf)[2]) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        float pv[16];
        // masked tiles: element i of this lane is key key0 + 4h + (i&3) + 8(i>>2); it counts while key <= klim: one
        // compare against a compile-time row offset per element
        const int rel = klim[qb_] - key0 - 4 * h;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            float x = sc[i];
            if constexpr (P == 2) x = fmaf(x, x, c0);
            if constexpr (MASKED) x = ((i & 3) + 8 * (i >> 2) <= rel) ? x : 0.f;
            pv[i] = x;
            gsum[qb_][i & 3] += x;
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const f32x4 x0 = {pv[8 * s], pv[8 * s + 1], pv[8 * s + 2], pv[8 * s + 3]};
            const f32x4 x1 = {pv[8 * s + 4], pv[8 * s + 5], pv[8 * s + 6], pv[8 * s + 7]};
            if constexpr (NPP == 2) {
                bf16x4 h0, l0, h1, l1;
                split4(x0, h0, l0);
                split4(x1, h1, l1);
                pf[s].p[0] = cat4(h0, h1);
                pf[s].p[1] = cat4(l0, l1);
            } else {
                pf[s].p[0] = cat4(to_bf16x4(x0), to_bf16x4(x1));
            }
        }
    };
    // S^T chain of one key half for query block qb_: u0 + K(half) Q^T
    auto qk_chain = [&](const Frag<NP> (&kf)[KS], int qb_) {
        f32x16 sc;
        if constexpr (UNIT) sc = mfma32_parts_c1<NP, NP>(kf[0], qf[qb_][0]);
        else sc = mfma32_parts<NP, NP>(kf[0], qf[qb_][0], cinit);
#pragma unroll
        for (int ks = 1; ks < KS; ++ks) sc = mfma32_parts<NP, NP>(kf[ks], qf[qb_][ks], sc);
        return sc;
    };
    constexpr int QKM = KS * (NP == 2 ? 3 : 1);                       // MFMAs of one S chain
    constexpr int PVM = 2 * DT * (1 + (NP == 2) + (NPP == 2));        // MFMAs of one key half of O^T += V^T P^T

    // One 64-key tile for one query block.  MASKED = the tile touches the causal diagonal or runs past N_k.
    // SCHED 1: software pipeline inside the wave (the two 32-key halves h0, h1 of the tile)
    //     A: S(h0) = K(h0) Q^T            B: S(h1) = K(h1) Q^T  ||  P(h0) = f(S(h0)) on the vector ALU
    //     C: O^T += V(h0)^T P(h0)^T  ||  P(h1) = f(S(h1))       D: O^T += V(h1)^T P(h1)^T
    // with B and C issued as {1 MFMA, a few VALU} groups; SCHED 0: both S chains, then per half the V^T fragments requested,
    // the polynomial walled off, the O^T product (compiler order otherwise)
    auto tile1 = [&](int kt, int stage, auto masked_tag) __attribute__((always_inline)) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        constexpr int VPOLY = (P == 2 ? 16 : 0) + 16 + (NPP == 2 ? 40 : 8) + (MASKED ? 32 : 0);   // VALU of one poly()
        const int KI = stage * STAGE, VI = KI + NP * KIMG, k0 = kt * 64;
        f32x16 sc[2];
        Frag<NP> kf[2][KS];
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int p = 0; p < NP; ++p) kf[jt][ks].p[p] = ld_row8<DP, 1>(smem, KI + p * KIMG, 32 * jt + l31, 2 * ks + h);
        Frag<NP> vf[2][2][DT];
        auto vread = [&](int jt) __attribute__((always_inline)) {
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int p = 0; p < NP; ++p) vf[jt][s][dt].p[p] = ld_tr8_32<DP>(smem, VI + p * VIMG, 32 * jt + 16 * s, 32 * dt, lane);
        };
        Frag<NPP> pf[2][2];
        auto pv = [&](int jt) __attribute__((always_inline)) {
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) oacc[0][dt] = mfma32_parts<NP, NPP>(vf[jt][s][dt], pf[jt][s], oacc[0][dt]);
        };
        if constexpr (SCHED == 1) {
            sc[0] = qk_chain(kf[0], 0);
            vread(0);
            __builtin_amdgcn_sched_barrier(0);
            sc[1] = qk_chain(kf[1], 0);
            poly(sc[0], 0, k0, masked_tag, pf[0]);
#pragma unroll
            for (int i = 0; i < QKM; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, (VPOLY + QKM - 1) / QKM, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            vread(1);
            pv(0);
            poly(sc[1], 0, k0 + 32, masked_tag, pf[1]);
#pragma unroll
            for (int i = 0; i < PVM; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
                __builtin_amdgcn_sched_group_barrier(0x002, (VPOLY + PVM - 1) / PVM, 1);
            }
            __builtin_amdgcn_sched_barrier(0);
            pv(1);
        } else {
            sc[0] = qk_chain(kf[0], 0);
            sc[1] = qk_chain(kf[1], 0);
#pragma unroll
            for (int jt = 0; jt < 2; ++jt) {
                vread(jt);
                __builtin_amdgcn_sched_barrier(0);
                poly(sc[jt], 0, k0 + 32 * jt, masked_tag, pf[jt]);
                __builtin_amdgcn_sched_barrier(0);
                pv(jt);
            }
        }
    };

    // One 64-key tile for TWO query blocks a, b (single-part operands).  Six phases; in every phase but the first and the
    // last the matrix instructions of one (block, key half) pair are issued beside the polynomial of another:
    //     1: S(a,h0) S(b,h0)     2: S(a,h1) || P(a,h0)     3: S(b,h1) || P(b,h0)
    //     4: O(a) += V(h0)^T P(a,h0) || P(a,h1)     5: O(b) += V(h0)^T P(b,h0) || P(b,h1)     6: O(a), O(b) += V(h1)^T P(.,h1)
    // K fragments of a key half serve both blocks, V^T fragments likewise.  SCHED 1 pins each phase's issue as
    // {1 MFMA, VPOLY / 4 VALU} groups; SCHED 0 only fences the phases.
    auto tile2 = [&](int kt, int stage, auto masked_tag) __attribute__((always_inline)) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        constexpr int VPOLY = (P == 2 ? 16 : 0) + 16 + 8 + (MASKED ? 32 : 0);
        const int KI = stage * STAGE, VI = KI + NP * KIMG, k0 = kt * 64;
        constexpr int B1 = QB - 1;                                   // second block (index 0 where this lambda is dead code)
        auto kread = [&](int jt, Frag<NP> (&kf)[KS]) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) kf[ks].p[0] = ld_row8<DP, 1>(smem, KI, 32 * jt + l31, 2 * ks + h);
        };
        auto vread = [&](int jt, Frag<NP> (&vf)[2][DT]) {
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) vf[s][dt].p[0] = ld_tr8_32<DP>(smem, VI, 32 * jt + 16 * s, 32 * dt, lane);
        };
        auto pv = [&](int qb_, const Frag<NP> (&vf)[2][DT], const Frag<NPP> (&pf)[2]) {
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) oacc[qb_][dt] = mfma32_parts<NP, NPP>(vf[s][dt], pf[s], oacc[qb_][dt]);
        };
        auto group = [&](auto id_tag) __attribute__((always_inline)) {
            constexpr int ID = decltype(id_tag)::value;
            if constexpr (SCHED == 1) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, ID);
                    __builtin_amdgcn_sched_group_barrier(0x002, (VPOLY + 3) / 4, ID);
                }
            }
        };
        Frag<NP> kf0[KS], kf1[KS], vf0[2][DT], vf1[2][DT];
        Frag<NPP> pa0[2], pb0[2], pa1[2], pb1[2];
        kread(0, kf0);
        kread(1, kf1);
        // 1
        f32x16 sa0 = qk_chain(kf0, 0);
        f32x16 sb0 = qk_chain(kf0, B1);
        vread(0, vf0);
        __builtin_amdgcn_sched_barrier(0);
        // 2
        f32x16 sa1 = qk_chain(kf1, 0);
        poly(sa0, 0, k0, masked_tag, pa0);
        group(std::integral_constant<int, 0>{});
        __builtin_amdgcn_sched_barrier(0);
        // 3
        f32x16 sb1 = qk_chain(kf1, B1);
        poly(sb0, B1, k0, masked_tag, pb0);
        group(std::integral_constant<int, 1>{});
        __builtin_amdgcn_sched_barrier(0);
        // 4
        vread(1, vf1);
        pv(0, vf0, pa0);
        poly(sa1, 0, k0 + 32, masked_tag, pa1);
        group(std::integral_constant<int, 2>{});
        __builtin_amdgcn_sched_barrier(0);
        // 5
        pv(B1, vf0, pb0);
        poly(sb1, B1, k0 + 32, masked_tag, pb1);
        group(std::integral_constant<int, 3>{});
        __builtin_amdgcn_sched_barrier(0);
        // 6
        pv(0, vf1, pa1);
        pv(B1, vf1, pb1);
    };
    auto tile = [&](int kt, int stage, auto masked_tag) __attribute__((always_inline)) {
        if constexpr (QB == 2) tile2(kt, stage, masked_tag);
        else tile1(kt, stage, masked_tag);
    };
    // staging half of an iteration: tile kt+1 goes to the other stage (last read before the previous barrier), tile kt+2
    // is requested
    auto advance = [&](int kt) __attribute__((always_inline)) {
        if (kt + 1 < nkt) {
            commit((kt & 1) ^ 1);
            if (kt + 2 < nkt) request(kt + 2);
        }
    };
    // per wave: tiles [0, n_plain) lie wholly below the diagonal and inside N_k, [n_plain, n_act) need the masks,
    // [n_act, nkt) lie wholly above the diagonal of this wave's queries (other waves of the workgroup still need them)
    const int n_full = Nk / 64;
    const int n_plain = causal ? min((qw0 + 1) / 64, n_full) : n_full;
    const int n_act = causal ? min(nkt, (qw0 + QW - 1) / 64 + 1) : nkt;

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

#pragma unroll
    for (int qb_ = 0; qb_ < QB; ++qb_) {
        const int myq = qw0 + 32 * qb_ + l31;
        float gs = (gsum[qb_][0] + gsum[qb_][1]) + (gsum[qb_][2] + gsum[qb_][3]);
        gs += __shfl_xor(gs, 32, 64);
        gs *= fscale;
        // unmasked: rowsum(f) carries the constant N_k; the reference's constant is g0 (fastmax.py:271, fastmax_hack.py:21)
        const float gval = causal ? gs : gs - (float)Nk + prm.g0;
        if (myq < Nq && prm.g && h == 0) prm.g[(int64_t)bh * Nq + myq] = gval;
        const float ginv = fscale / gval;
        store_tile32_t<DT>(smem + w * 4096, oacc[qb_], ginv, lane, prm.o, prm.out_dtype, (int64_t)bh * Nq, qw0 + 32 * qb_, Nq, D);
    }
}

template <int DP, int P, typename TIN, int NPP, int NW, bool UNIT, int SCHED, int QB>
static int launch_quad32_w(Quad32Params prm, hipStream_t stream) {
    constexpr int NP = InTraits<TIN>::NP;
    constexpr int stages = 2 * NP * (img_bytes<DP, 1>() + img_bytes<DP, 2>()), epi = NW * 4096;
    constexpr int lds = stages > epi ? stages : epi;
    auto kern = fwd_quad32_kernel<DP, P, TIN, NPP, NW, UNIT, SCHED, QB>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    prm.nqt = (prm.Nq + 32 * QB * NW - 1) / (32 * QB * NW);
    hipLaunchKernelGGL(kern, dim3(prm.nqt * prm.BH), dim3(64 * NW), lds, stream, prm);
    return (int)hipGetLastError();
}
template <int DP, int P, typename TIN, int NPP, bool UNIT>
static int launch_quad32_n(const Quad32Params& prm, hipStream_t stream) {
    static const int sched = [] { const char* e = getenv("FASTMAX_QUAD32_SCHED"); return e ? atoi(e) : -1; }();
    static const int qbenv = [] { const char* e = getenv("FASTMAX_QUAD32_QB"); return e ? atoi(e) : 0; }();
    constexpr bool SINGLE = InTraits<TIN>::NP == 1 && NPP == 1;
    if constexpr (SINGLE && DP == 64) {
        // bf16, D <= 64, 16-bit result: two query blocks per wave once a head has enough query tiles of 256
        const int qb = qbenv ? qbenv : (prm.Nq >= 1024 ? 2 : 1);
        if (qb == 2) {
            if (sched == 0) return launch_quad32_w<DP, P, TIN, NPP, 4, UNIT, 0, 2>(prm, stream);
            return launch_quad32_w<DP, P, TIN, NPP, 4, UNIT, 1, 2>(prm, stream);
        }
        if (sched == 1) return launch_quad32_w<DP, P, TIN, NPP, 4, UNIT, 1, 1>(prm, stream);
        return launch_quad32_w<DP, P, TIN, NPP, 4, UNIT, 0, 1>(prm, stream);
    } else {
        // everything else: one query block per wave; eight waves unless the operands are bf16 at D <= 64; D = 128 with the
        // grouped issue order
        constexpr int NWD = (DP == 64 && InTraits<TIN>::NP == 1) ? 4 : 8;
        constexpr int DEF_SCHED = DP == 128 ? 1 : 0;
        if ((sched >= 0 ? sched : DEF_SCHED) == 1) return launch_quad32_w<DP, P, TIN, NPP, NWD, UNIT, 1, 1>(prm, stream);
        return launch_quad32_w<DP, P, TIN, NPP, NWD, UNIT, 0, 1>(prm, stream);
    }
}
// a = 2^e exactly: scaling single-part (bf16) query fragments by it loses nothing
static bool is_pow2(float a) { int e; return a > 0.f && frexpf(a, &e) == 0.5f; }
template <int DP, int P, typename TIN, int NPP>
static int launch_quad32_u(const Quad32Params& prm, hipStream_t stream) {
    if constexpr (InTraits<TIN>::NP == 2) return launch_quad32_n<DP, P, TIN, NPP, true>(prm, stream);
    else return is_pow2(prm.a) ? launch_quad32_n<DP, P, TIN, NPP, true>(prm, stream) : launch_quad32_n<DP, P, TIN, NPP, false>(prm, stream);
}
template <int DP, int P, typename TIN>
static int launch_quad32_t(const Quad32Params& prm, hipStream_t stream) {
    if constexpr (InTraits<TIN>::NP == 1) {
        if (prm.out_dtype != FASTMAX_F32) return launch_quad32_u<DP, P, TIN, 1>(prm, stream);
    }
    return launch_quad32_u<DP, P, TIN, 2>(prm, stream);
}
template <int P, typename TIN>
static int launch_quad32_d(const Quad32Params& prm, hipStream_t stream) {
    if constexpr (InTraits<TIN>::NP == 2) {          // two-part operands stop at D = 64 (quad32_supported)
        return prm.D <= 64 ? launch_quad32_t<64, P, TIN>(prm, stream) : FASTMAX_E_BAD_SHAPE;
    } else {
        return prm.D <= 64 ? launch_quad32_t<64, P, TIN>(prm, stream) : launch_quad32_t<128, P, TIN>(prm, stream);
    }
}
template <typename TIN>
static int launch_quad32_p(const Quad32Params& prm, int p, hipStream_t stream) {
    return p == 1 ? launch_quad32_d<1, TIN>(prm, stream) : launch_quad32_d<2, TIN>(prm, stream);
}

// worth it once a head has a few hundred queries; shorter problems stay on the 16-query-per-wave kernel
bool quad32_supported(const fastmax_problem& p) {
    static const int mode = [] { const char* e = getenv("FASTMAX_QUAD32"); return e ? atoi(e) : 1; }();
    if (!mode) return false;
    const int epl = p.in_dtype == FASTMAX_F32 ? 4 : 8;
    // two-part operands (fp32 / fp16) at D > 64 do not fit the register file in this decomposition (measured 4.3 vs 2.5 ms):
    // they stay on the 16-query-per-wave kernel
    if (p.in_dtype != FASTMAX_BF16 && p.D > 64) return false;
    return (p.D % epl) == 0 && p.D <= 128 && p.Nq >= 256 && (int64_t)p.B * p.H * ((p.Nq + 127) / 128) <= 0x7fffffff;
}

int launch_fwd_quad32(const FwdArgs& a) {
    if (!quad32_supported(a.prob)) return FASTMAX_E_BAD_SHAPE;
    const int eb = a.prob.in_dtype == FASTMAX_F32 ? 4 : 2;
    if (!quad32_span_ok(a.ks.sn, a.prob.Nk, a.prob.D, eb) || !quad32_span_ok(a.vs.sn, a.prob.Nk, a.prob.D, eb)) return launch_fwd_quad_mfma(a);
    Quad32Params prm{a.q, a.k, a.v, a.qs, a.ks, a.vs, a.o, a.g, a.prob.H, a.prob.B * a.prob.H, a.prob.Nq, a.prob.Nk, a.prob.D,
                     a.prob.causal, a.prob.out_dtype, 0, a.prob.a, a.prob.g0};
    switch (a.prob.in_dtype) {
        case FASTMAX_F32: return launch_quad32_p<float>(prm, a.prob.p, a.stream);
        case FASTMAX_BF16: return launch_quad32_p<bf16_t>(prm, a.prob.p, a.stream);
        case FASTMAX_F16: return launch_quad32_p<f16_t>(prm, a.prob.p, a.stream);
    }
    return FASTMAX_E_BAD_DTYPE;
}

}  // namespace fastmax
