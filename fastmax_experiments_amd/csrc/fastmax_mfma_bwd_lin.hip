// fastmax p=1 masked BACKWARD in linear time on the matrix cores (gfx950), D <= 64.
//
// Hand-derived gradients of attention_mechanisms/fastmax.py:432-485 (dQ: forward prefix sums),
// 541-604 (dK) and 647-691 (dV: reverse cumulative sums) without the (N,D,D) temporaries.  With
// w_i = 1/g_i, c_i = G_i.o_i, ghat_i = w_i G_i, e_i = -w_i c_i and T_ij = ghat_i.v_j + e_i (j <= i):
//   dQ_i = a [ S2_i ghat_i + ksum_i e_i ]            S2_i = sum_{j<=i} k_j v_j^T,  ksum_i = sum_{j<=i} k_j
//   dK_j = a [ R2_j v_j + rq_j ]                     R2_j = sum_{i>=j} q_i ghat_i^T, rq_j = sum_{i>=j} q_i e_i
//   dV_j = R1_j + a R2_j^T k_j                       R1_j = sum_{i>=j} ghat_i
// evaluated per 64-token chunk as (carried state) + (masked 64x64 tile), exactly like the forward kernel:
//   bwd_p1_dq_kernel    walks the chunks forwards  (state S2 in MFMA accumulators, image [m][d] in LDS)
//   bwd_p1_dkv_kernel   walks the chunks backwards (state R2 likewise; R1, rq exact fp32 vectors)
// The dQ kernel also writes c_i to the workspace for the dK/dV kernel.
#include "fastmax_mfma_common.h"

namespace fastmax {

struct LinBwdParams {
    const void *q, *k, *v, *o, *go;
    const float* g;
    Strides3 qs, ks, vs, gos;
    void *dq, *dk, *dv;
    float* c;                       // workspace (B,H,N)
    int H, N, D, grad_dtype, o_dtype;
    float a;
    // sequence split for few heads (fastmax_mfma_split.hip): segment s of the dQ scan starts from fstate record s-1
    // (sums over the earlier segments), segment s of the dK/dV scan from rstate record s (sums over the later ones)
    const float *fstate, *rstate;
    int nseg, cps;
    // NORM (linearmax training route): q, k are raw; the prologue (x - mean) * scale[bh] is applied while staging
    const float *qscale, *kscale;
    // FUSEK (with NORM): dk leaves as the gradient wrt the RAW k -- the prologue's row-wise backward inv (g - mean_D g) is applied
    // to the dK tile before it is stored, and every block leaves sum_n g_n . xc_n for the one-row fix-up (fastmax_normalize.hip),
    // [bh][nseg] entries
    float* kpart_dot;
    const int* k_nstar;             // the row of k that attains the max-norm, per head (found by the forward's statistics)
    const int* q_nstar;             // FUSEQ: the same for q (dq leaves as the gradient wrt the raw q)
};

// Stage a wave's 16 x W fp32 accumulator tile (lane = row r, acc[t][reg] = column 16t + 4q4 + reg) through a wave-private
// 16*W*4-byte LDS area and write it out as whole row segments at columns [col0, col0 + W) of the output rows.
template <int W>
__device__ __forceinline__ void store_cols16(char* ost, const f32x4 (&acc)[W / 16], float scale, int lane, void* out, int dtype,
                                             int64_t row0_elem, int col0, int first_row, int nrows, int D) {
    constexpr int C16 = W / 4, T = W / 16;
    const int r = lane & 15, q4 = lane >> 4;
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int c16 = 4 * t + q4;
        *reinterpret_cast<f32x4*>(ost + r * (W * 4) + (((c16 ^ r) & (C16 - 1)) << 4)) = acc[t] * scale;
    }
#pragma unroll
    for (int u = 0; u < (16 * C16) / 64; ++u) {
        const int idx = u * 64 + lane, rl = idx / C16, c16 = idx % C16;
        const f32x4 val = *reinterpret_cast<const f32x4*>(ost + rl * (W * 4) + (((c16 ^ rl) & (C16 - 1)) << 4));
        if (first_row + rl < nrows && col0 + 4 * c16 < D)
            store4_any(out, dtype, row0_elem + (int64_t)rl * D + col0 + 4 * c16, val);
    }
}

// State images (S2 for dQ, a R2 for dK/dV): hi + lo bf16 parts for fp32 / fp16 problems, ONE rounded part for bf16
// problems (SP = 1) -- as in the forward kernel its rounding is far below the bf16 rounding of the gradients.
template <int DP, int SP, int TW> __device__ __forceinline__ void publish_state(char* smem, int base, int simg, const f32x4 (&acc)[TW],
                                                                  float scale, int row, int q4, int t0) {
    // accumulators (rows = 16(t0+t) + 4q4 + reg, column = `row` on the lane) -> bf16 image row `row`
    // (`row` is made opaque: the TW store addresses are loop invariants the compiler otherwise hoists out of the chunk loop and,
    // in the kernels that use every register, spills -- each reload then sits behind an s_waitcnt vmcnt(0))
    asm volatile("" : "+v"(row));
#pragma unroll
    for (int tt = 0; tt < TW; ++tt) {
        const int t = t0 + tt;
        const int off = img_off<DP>(row, 2 * t + (q4 >> 1)) + ((q4 & 1) << 3);
        if constexpr (SP == 1) {
            *reinterpret_cast<bf16x4*>(smem + base + off) = to_bf16x4(acc[tt] * scale);
        } else {
            bf16x4 hi, lo;
            split4(acc[tt] * scale, hi, lo);
            *reinterpret_cast<bf16x4*>(smem + base + off) = hi;
            *reinterpret_cast<bf16x4*>(smem + base + simg + off) = lo;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// dQ: grid = B*H, block = 256
// ------------------------------------------------------------------------------------------------
// NW = 4: wave w owns query tile w and every output column.  NW = 8 (the variants whose LDS footprint allows one workgroup
// per CU only): wave (wq = w & 3, hf = w >> 2) owns query tile wq and the column half hf of dQ and of the S2 state; the
// 64 x 64 score-shaped tile is computed by both halves.
template <int DP, typename TIN, int NW, bool NORM, bool BUF, bool FUSEQ = false>
__global__ __launch_bounds__(64 * NW, (DP == 64 && NW == 4) ? 2 : 1) void bwd_p1_dq_kernel(LinBwdParams prm) {
    static_assert(!FUSEQ || NORM, "the fused prologue backward belongs to the linearmax route");
    constexpr int NP = InTraits<TIN>::NP, EPL = InTraits<TIN>::EPL;
    constexpr int C = 64, IMG = C * DP * 2, SIMG = DP * DP * 2, SP = NP;
    constexpr int KI = 0, VI = NP * IMG, GI = 2 * NP * IMG, S2I = 3 * NP * IMG;
    constexpr int KSUM = S2I + SP * SIMG;
    constexpr int NT = 64 * NW, HF = NW / 4;
    constexpr int COLS = DP / EPL, RPP = NT / COLS, NPASS = C / RPP;
    constexpr int PARTK = KSUM + 2 * DP * 4, CS = PARTK + RPP * DP * 4, WS = CS + 256, OST = WS + 256;
    constexpr int RSQ = OST + (NW == 8 ? 8 * 16 * (DP / 2) * 4 : 0);           // FUSEQ: NW x 16 row sums of the dQ tile
    constexpr int KS = DP / 32, NSL = DP / 64;     // NSL state row slabs (16 rows) per wave
    constexpr int MT = (DP / 16) / HF, DT = (DP / 16) / HF;                  // column tiles of this wave (dQ columns m, S2 columns d)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wq = w & 3, hf = w >> 2, t0 = hf * MT;
    const int r = lane & 15, q4 = lane >> 4;
    const int bh = blockIdx.x / prm.nseg, seg = blockIdx.x - bh * prm.nseg, b = bh / prm.H, h = bh % prm.H;
    const int N = prm.N, D = prm.D;
    const TIN* kb = reinterpret_cast<const TIN*>(prm.k) + (int64_t)b * prm.ks.sb + (int64_t)h * prm.ks.sh;
    const TIN* vb = reinterpret_cast<const TIN*>(prm.v) + (int64_t)b * prm.vs.sb + (int64_t)h * prm.vs.sh;
    const TIN* gb = reinterpret_cast<const TIN*>(prm.go) + (int64_t)b * prm.gos.sb + (int64_t)h * prm.gos.sh;
    const int srow = tid / COLS, scol = tid % COLS;
    float ksc_c = 0.f;
    if constexpr (NORM) ksc_c = scol * EPL < D ? prm.kscale[bh] : 0.f;
    const float invD = 1.0f / (float)D;
    float qinv = 1.f;                                            // FUSEQ: the head's 1 / max-norm of q, a scalar read once
    if constexpr (FUSEQ) qinv = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, prm.qscale[bh])));

    // o is contiguous (B,H,N,D) in the input dtype on the masked path (dtype rule Q1)
    const TIN* ob = reinterpret_cast<const TIN*>(prm.o) + (int64_t)bh * N * D;
    u32x4 rk[NPASS], rv[NPASS], rg[NPASS], ro[NPASS];
    float rw[NPASS];
    const ScanLoader<BUF, TIN, NPASS, RPP> kload(kb, prm.ks.sn, N, D, DP, srow, scol), vload(vb, prm.vs.sn, N, D, DP, srow, scol),
        gload(gb, prm.gos.sn, N, D, DP, srow, scol), oload(ob, D, N, D, DP, srow, scol);
    auto issue = [&](int n0) {
        kload.load(n0 / C, rk);
        vload.load(n0 / C, rv);
        gload.load(n0 / C, rg);
        oload.load(n0 / C, ro);
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int row = n0 + srow + ps * RPP;
            const int rc = row < N ? row : N - 1;
            // the raw g_i: its reciprocal is taken where it is consumed -- taken here it would need the value at once, i.e.
            // an s_waitcnt vmcnt(0) right behind the tile loads (the whole prefetch waited on at issue time)
            rw[ps] = prm.g[(int64_t)bh * N + rc];
        }
    };
    for (int i = tid; i < (SP * SIMG) / 16; i += NT) *reinterpret_cast<f32x4*>(smem + S2I + 16 * i) = f32x4{0, 0, 0, 0};
    if (tid < DP) reinterpret_cast<float*>(smem + KSUM)[tid] = 0.f;
    f32x4 s2acc[NSL][DT];                                        // S2[m = 16(w + 4sl) + r][d = 16dt + 4q4 + reg]
#pragma unroll
    for (int sl = 0; sl < NSL; ++sl)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) s2acc[sl][dt] = f32x4{0, 0, 0, 0};

    const int nchunks = (N + C - 1) / C;
    const int c_begin = seg * prm.cps, c_end = min(nchunks, c_begin + prm.cps);
    if (seg > 0) {
        __syncthreads();                                             // zero fill done before the images are overwritten
        const float* rec = prm.fstate + ((int64_t)bh * (prm.nseg - 1) + (seg - 1)) * (DP * DP + 2 * DP);
#pragma unroll
        for (int sl = 0; sl < NSL; ++sl) {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
                s2acc[sl][dt] = *reinterpret_cast<const f32x4*>(rec + (16 * (wq + 4 * sl) + r) * DP + 16 * (t0 + dt) + 4 * q4);
            publish_state<DP, SP, DT>(smem, S2I, SIMG, s2acc[sl], 1.0f, 16 * (wq + 4 * sl) + r, q4, t0);
        }
        if (tid < DP) reinterpret_cast<float*>(smem + KSUM)[DP * (c_begin & 1) + tid] = rec[DP * DP + DP + tid];
    }
    issue(c_begin * C);
    __syncthreads();
    for (int c = c_begin; c < c_end; ++c) {
        const int n0 = c * C, cur = c & 1, nxt = cur ^ 1;
        const float* ksum_cur = reinterpret_cast<const float*>(smem + KSUM) + DP * cur;
        {
            float ck[EPL];
#pragma unroll
            for (int e = 0; e < EPL; ++e) ck[e] = 0.f;
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps) {
                const int row = srow + ps * RPP;
                float xk[EPL], xg[EPL], xo[EPL];
                piece_to_float<TIN>(rk[ps], xk);
                piece_to_float<TIN>(rg[ps], xg);
                piece_to_float<TIN>(ro[ps], xo);
                if constexpr (NORM) {
                    normalize_piece<COLS, EPL>(xk, ksc_c, invD);
                    stage_floats<DP, EPL, NP>(smem, KI, row, scol, xk);
                } else {
                    stage_piece<DP, TIN>(smem, KI, row, scol, rk[ps]);
                }
                stage_piece<DP, TIN>(smem, VI, row, scol, rv[ps]);
                stage_piece<DP, TIN>(smem, GI, row, scol, rg[ps]);
                float part = 0.f;
#pragma unroll
                for (int e = 0; e < EPL; ++e) { part = fmaf(xg[e], xo[e], part); ck[e] += xk[e]; }
                part = rowgroup_sum<COLS>(part);                 // c_i = G_i . o_i
                if (scol == COLS - 1) {
                    reinterpret_cast<float*>(smem + CS)[row] = part;
                    reinterpret_cast<float*>(smem + WS)[row] = n0 + row < N ? 1.0f / rw[ps] : 0.f;
                    if (n0 + row < N) prm.c[(int64_t)bh * N + n0 + row] = part;
                }
            }
#pragma unroll
            for (int e = 0; e < EPL; ++e) reinterpret_cast<float*>(smem + PARTK)[srow * DP + scol * EPL + e] = ck[e];
        }
        if (c + 1 < c_end) issue(n0 + C);
        __syncthreads();                                             // B1
        if (tid < DP) {
            float s = ksum_cur[tid];
#pragma unroll 8
            for (int g16 = 0; g16 < RPP; ++g16) s += reinterpret_cast<const float*>(smem + PARTK)[g16 * DP + tid];
            reinterpret_cast<float*>(smem + KSUM)[DP * nxt + tid] = s;
        }
        // ---- phase A: dQ^T[m][i] for this wave's 16 queries ------------------------------------------
        const int qi = 16 * wq + r;
        const float ci = reinterpret_cast<const float*>(smem + CS)[qi];
        const float wi = reinterpret_cast<const float*>(smem + WS)[qi];
        Frag<NP> gf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int p = 0; p < NP; ++p) gf[ks].p[p] = ld_row8<DP>(smem, GI + p * IMG, qi, 4 * ks + q4);
        f32x4 acc[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            acc[mt] = *reinterpret_cast<const f32x4*>(ksum_cur + 16 * (t0 + mt) + 4 * q4) * (-ci);     // ksum_prev * (-c_i)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                Frag<SP> sf;
#pragma unroll
                for (int p = 0; p < SP; ++p) sf.p[p] = *reinterpret_cast<const bf16x8*>(smem + S2I + p * SIMG + img_off<DP>(16 * (t0 + mt) + r, 4 * ks + q4));
                acc[mt] = mfma_parts<SP, NP>(sf, gf[ks], acc[mt]);                                  // S2_prev G_i
            }
        }
        Frag<2> tf[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            f32x4 tt[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int jt = 2 * s + e;
                f32x4 u = {-ci, -ci, -ci, -ci};
                if (jt <= wq) {
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        Frag<NP> vf;
#pragma unroll
                        for (int p = 0; p < NP; ++p) vf.p[p] = ld_row8<DP>(smem, VI + p * IMG, 16 * jt + r, 4 * ks + q4);
                        u = mfma_parts<NP, NP>(vf, gf[ks], u);                                     // v_j . G_i - c_i
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bool keep = (jt < wq) || (jt == wq && (4 * q4 + i) <= r);
                    tt[e][i] = keep ? u[i] : 0.f;
                }
            }
            bf16x4 h0, l0, h1, l1;
            split4(tt[0], h0, l0);
            split4(tt[1], h1, l1);
            tf[s].p[0] = cat4(h0, h1);
            tf[s].p[1] = cat4(l0, l1);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (2 * s <= wq) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    Frag<NP> ktf;
#pragma unroll
                    for (int p = 0; p < NP; ++p) ktf.p[p] = ld_tr8<DP>(smem, KI + p * IMG, 32 * s, 16 * (t0 + mt), lane);
                    acc[mt] = mfma_parts<NP, 2>(ktf, tf[s], acc[mt]);
                }
            }
        }
        // ---- phase B: S2[m = 16w + r][d] += sum_j K[j][m] V[j][d]  (rows d in registers) --------------
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            Frag<NP> kf[NSL];
#pragma unroll
            for (int sl = 0; sl < NSL; ++sl)
#pragma unroll
                for (int p = 0; p < NP; ++p) kf[sl].p[p] = ld_tr8<DP>(smem, KI + p * IMG, 32 * s, 16 * (wq + 4 * sl), lane);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                Frag<NP> vtf;
#pragma unroll
                for (int p = 0; p < NP; ++p) vtf.p[p] = ld_tr8<DP>(smem, VI + p * IMG, 32 * s, 16 * (t0 + dt), lane);
#pragma unroll
                for (int sl = 0; sl < NSL; ++sl) s2acc[sl][dt] = mfma_parts<NP, NP>(vtf, kf[sl], s2acc[sl][dt]);
            }
        }
        // FUSEQ: the prologue's row-wise backward on the dQ tile -- dq_raw = inv_q (dq' - mean_D dq'), dq' = a w_i acc (the dL/dM
        // term of row n* is added by the fix-up pass).  The row sum of a lane's columns, across q4 by shuffles, across the two
        // column halves (NW = 8) through LDS and B2: the store then happens after B2 (its staging area is wave-private).
        float qrs = 0.f, oscale = prm.a * wi;
        if constexpr (FUSEQ) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int i = 0; i < 4; ++i) qrs += acc[mt][i];
            qrs += __shfl_xor(qrs, 16, 64);
            qrs += __shfl_xor(qrs, 32, 64);
            oscale *= qinv;
            if constexpr (NW == 8) {
                if (q4 == 0) reinterpret_cast<float*>(smem + RSQ)[16 * w + r] = qrs;
            } else {
                const float qmean = qrs * invD;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[mt][i] -= qmean;                    // padded columns are not stored
            }
        }
        if constexpr (NW == 4) {
            // dQ rows, staged in the gradient dtype through this wave's own (already consumed) G image rows
            store_tile16_private<DP, sizeof(TIN)>(smem + GI + 16 * w * (2 * DP), smem + GI + IMG + 16 * w * (2 * DP), acc, oscale,
                                                  lane, prm.dq, prm.grad_dtype, ((int64_t)bh * N + n0 + 16 * w) * D, n0 + 16 * w, N, D);
        } else if constexpr (!FUSEQ) {
            store_cols16<16 * MT>(smem + OST + w * (16 * 16 * MT * 4), acc, oscale, lane, prm.dq, prm.grad_dtype,
                                  ((int64_t)bh * N + n0 + 16 * wq) * D, 16 * t0, n0 + 16 * wq, N, D);
        }
        __syncthreads();                                             // B2
        if constexpr (FUSEQ && NW == 8) {
            const float qmean = (qrs + reinterpret_cast<const float*>(smem + RSQ)[16 * (w ^ 4) + r]) * invD;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[mt][i] -= qmean;
            store_cols16<16 * MT>(smem + OST + w * (16 * 16 * MT * 4), acc, oscale, lane, prm.dq, prm.grad_dtype,
                                  ((int64_t)bh * N + n0 + 16 * wq) * D, 16 * t0, n0 + 16 * wq, N, D);
        }
        if (c + 1 < c_end) {
#pragma unroll
            for (int sl = 0; sl < NSL; ++sl) publish_state<DP, SP, DT>(smem, S2I, SIMG, s2acc[sl], 1.0f, 16 * (wq + 4 * sl) + r, q4, t0);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// dK, dV: grid = B*H, block = 256; chunks are walked from the last to the first
// ------------------------------------------------------------------------------------------------
// NW = 8: wave (wk = w & 3, hf = w >> 2) owns the 16 keys of tile wk and the column half hf of dK, dV and of the R2 state
// FUSEK: 0 none; 1 the prologue's backward for k on the dK tile + the shared sum T; 2 the sum T only (grouped-query heads: dk' is
// summed over the group by the prologue's own backward pass, but the q side is finished in the dQ kernel and needs T)
template <int DP, typename TIN, int NW, bool NORM, bool BUF, int FUSEK = 0>
__global__ __launch_bounds__(64 * NW, (InTraits<TIN>::NP == 1 && DP == 64 && NW == 4) ? 2 : 1) void bwd_p1_dkv_kernel(LinBwdParams prm) {
    static_assert(!FUSEK || NORM, "the fused prologue backward belongs to the linearmax route");
    constexpr int NP = InTraits<TIN>::NP, EPL = InTraits<TIN>::EPL;
    constexpr int C = 64, IMG = C * DP * 2, SIMG = DP * DP * 2, SP = NP;
    constexpr int QI = 0, KI = NP * IMG, VI = 2 * NP * IMG, GI = 3 * NP * IMG, R2I = 4 * NP * IMG;
    constexpr int R1 = R2I + SP * SIMG, RQ = R1 + 2 * DP * 4;
    constexpr int NT = 64 * NW, HF = NW / 4;
    constexpr int COLS = DP / EPL, RPP = NT / COLS, NPASS = C / RPP;
    constexpr int PARTG = RQ + 2 * DP * 4, PARTQ = PARTG + RPP * DP * 4, ES = PARTQ + RPP * DP * 4;
    constexpr int RSUM = ES + 256;                                           // FUSEK: NW x 16 row sums of the dK tile (+ the final reduce)
    constexpr int KS = DP / 32, NSL = DP / 64;
    constexpr int MT = (DP / 16) / HF, DT = (DP / 16) / HF;                  // column tiles of this wave
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wk = w & 3, hf = w >> 2, t0 = hf * MT;
    const int r = lane & 15, q4 = lane >> 4;
    const int bh = blockIdx.x / prm.nseg, seg = blockIdx.x - bh * prm.nseg, b = bh / prm.H, h = bh % prm.H;
    const int N = prm.N, D = prm.D;
    const float a = prm.a;
    const TIN* qb = reinterpret_cast<const TIN*>(prm.q) + (int64_t)b * prm.qs.sb + (int64_t)h * prm.qs.sh;
    const TIN* kb = reinterpret_cast<const TIN*>(prm.k) + (int64_t)b * prm.ks.sb + (int64_t)h * prm.ks.sh;
    const TIN* vb = reinterpret_cast<const TIN*>(prm.v) + (int64_t)b * prm.vs.sb + (int64_t)h * prm.vs.sh;
    const TIN* gb = reinterpret_cast<const TIN*>(prm.go) + (int64_t)b * prm.gos.sb + (int64_t)h * prm.gos.sh;
    const int srow = tid / COLS, scol = tid % COLS;
    float qsc_c = 0.f, ksc_c = 0.f;
    if constexpr (NORM) {
        qsc_c = scol * EPL < D ? prm.qscale[bh] : 0.f;
        ksc_c = scol * EPL < D ? prm.kscale[bh] : 0.f;
    }
    const float invD = 1.0f / (float)D;

    u32x4 rq[NPASS], rk[NPASS], rv[NPASS], rg[NPASS];
    float rw[NPASS], rc[NPASS];
    const ScanLoader<BUF, TIN, NPASS, RPP> qload(qb, prm.qs.sn, N, D, DP, srow, scol), kload(kb, prm.ks.sn, N, D, DP, srow, scol),
        vload(vb, prm.vs.sn, N, D, DP, srow, scol), gload(gb, prm.gos.sn, N, D, DP, srow, scol);
    auto issue = [&](int n0) {
        qload.load(n0 / C, rq);
        kload.load(n0 / C, rk);
        vload.load(n0 / C, rv);
        gload.load(n0 / C, rg);
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int row = n0 + srow + ps * RPP;
            const int rcl = row < N ? row : N - 1;
            rw[ps] = prm.g[(int64_t)bh * N + rcl];                   // raw g_i, see the dQ kernel
            rc[ps] = prm.c[(int64_t)bh * N + rcl];
        }
    };
    for (int i = tid; i < (SP * SIMG) / 16; i += NT) *reinterpret_cast<f32x4*>(smem + R2I + 16 * i) = f32x4{0, 0, 0, 0};
    if (tid < DP) {
        reinterpret_cast<float*>(smem + R1)[tid] = 0.f;          // parity buffers are indexed by (chunk & 1)
        reinterpret_cast<float*>(smem + R1)[DP + tid] = 0.f;
        reinterpret_cast<float*>(smem + RQ)[tid] = 0.f;
        reinterpret_cast<float*>(smem + RQ)[DP + tid] = 0.f;
    }
    f32x4 r2acc[NSL][DT];                                        // R2[m = 16(w + 4sl) + r][d = 16dt + 4q4 + reg]
#pragma unroll
    for (int sl = 0; sl < NSL; ++sl)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) r2acc[sl][dt] = f32x4{0, 0, 0, 0};

    float kdot = 0.f;                                            // FUSEK: this thread's share of sum_n dk'_n . y_n
    const int nchunks = (N + C - 1) / C;
    const int c_begin = seg * prm.cps, c_end = min(nchunks, c_begin + prm.cps);
    if (seg + 1 < prm.nseg) {
        __syncthreads();
        const float* rec = prm.rstate + ((int64_t)bh * (prm.nseg - 1) + seg) * (DP * DP + 2 * DP);
#pragma unroll
        for (int sl = 0; sl < NSL; ++sl) {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
                r2acc[sl][dt] = *reinterpret_cast<const f32x4*>(rec + (16 * (wk + 4 * sl) + r) * DP + 16 * (t0 + dt) + 4 * q4);
            publish_state<DP, SP, DT>(smem, R2I, SIMG, r2acc[sl], a, 16 * (wk + 4 * sl) + r, q4, t0);
        }
        if (tid < DP) {
            const int par = (c_end - 1) & 1;
            reinterpret_cast<float*>(smem + R1)[DP * par + tid] = rec[DP * DP + tid];
            reinterpret_cast<float*>(smem + RQ)[DP * par + tid] = a * rec[DP * DP + DP + tid];
        }
    }
    issue((c_end - 1) * C);
    __syncthreads();
    for (int c = c_end - 1; c >= c_begin; --c) {
        const int n0 = c * C, cur = c & 1, nxt = cur ^ 1;
        const float* r1_cur = reinterpret_cast<const float*>(smem + R1) + DP * cur;
        const float* rq_cur = reinterpret_cast<const float*>(smem + RQ) + DP * cur;
        {
            float cg[EPL], cq[EPL];
#pragma unroll
            for (int e = 0; e < EPL; ++e) { cg[e] = 0.f; cq[e] = 0.f; }
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps) {
                const int row = srow + ps * RPP;
                float xq[EPL], xg[EPL];
                piece_to_float<TIN>(rq[ps], xq);
                piece_to_float<TIN>(rg[ps], xg);
                const float wi = n0 + row < N ? 1.0f / rw[ps] : 0.f, ei = -wi * rc[ps];
                if constexpr (NORM) {
                    float xk[EPL];
                    piece_to_float<TIN>(rk[ps], xk);
                    normalize_piece<COLS, EPL>(xq, qsc_c, invD);
                    normalize_piece<COLS, EPL>(xk, ksc_c, invD);
                    stage_floats<DP, EPL, NP>(smem, QI, row, scol, xq);
                    stage_floats<DP, EPL, NP>(smem, KI, row, scol, xk);
                } else {
                    stage_piece<DP, TIN>(smem, QI, row, scol, rq[ps]);
                    stage_piece<DP, TIN>(smem, KI, row, scol, rk[ps]);
                }
#pragma unroll
                for (int e = 0; e < EPL; ++e) { xg[e] *= wi; cg[e] += xg[e]; cq[e] = fmaf(xq[e], ei, cq[e]); }
                stage_piece<DP, TIN>(smem, VI, row, scol, rv[ps]);
                stage_floats<DP, EPL, NP>(smem, GI, row, scol, xg);                  // ghat = w G
                if (scol == 0) reinterpret_cast<float*>(smem + ES)[row] = ei;
            }
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                reinterpret_cast<float*>(smem + PARTG)[srow * DP + scol * EPL + e] = cg[e];
                reinterpret_cast<float*>(smem + PARTQ)[srow * DP + scol * EPL + e] = cq[e];
            }
        }
        if (c > c_begin) issue(n0 - C);
        __syncthreads();                                             // B1
        for (int t = tid; t < 2 * DP; t += NT) {                    // suffix sums for the next (earlier) chunk
            const int col = t % DP;
            const bool isg = t < DP;
            const float* part = reinterpret_cast<const float*>(smem + (isg ? PARTG : PARTQ));
            float* base = reinterpret_cast<float*>(smem + (isg ? R1 : RQ));
            float s = 0.f;
#pragma unroll 8
            for (int g16 = 0; g16 < RPP; ++g16) s += part[g16 * DP + col];
            base[DP * nxt + col] = base[DP * cur + col] + (isg ? s : a * s);
        }
        // ---- phase A: this wave's 16 keys j = 16w + r ---------------------------------------------------
        const int kj = 16 * wk + r;
        Frag<NP> kf[KS], vf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int p = 0; p < NP; ++p) vf[ks].p[p] = ld_row8<DP>(smem, VI + p * IMG, kj, 4 * ks + q4);
        f32x4 dkacc[MT], dvacc[DT];
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            // dK^T[m][j] = rq'[m] + sum_d (a R2)[m][d] V[j][d]
            dkacc[t] = *reinterpret_cast<const f32x4*>(rq_cur + 16 * (t0 + t) + 4 * q4);
            // dV^T[d][j] = R1[d] + sum_m (a R2)[m][d] K[j][m]
            dvacc[t] = *reinterpret_cast<const f32x4*>(r1_cur + 16 * (t0 + t) + 4 * q4);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                Frag<SP> rf, rtf;
#pragma unroll
                for (int p = 0; p < SP; ++p) {
                    rf.p[p] = *reinterpret_cast<const bf16x8*>(smem + R2I + p * SIMG + img_off<DP>(16 * (t0 + t) + r, 4 * ks + q4));
                    rtf.p[p] = ld_tr8<DP>(smem, R2I + p * SIMG, 32 * ks, 16 * (t0 + t), lane);     // A[row d][k = m], permuted k
                }
                dkacc[t] = mfma_parts<SP, NP>(rf, vf[ks], dkacc[t]);
                Frag<NP> kpf;
#pragma unroll
                for (int p = 0; p < NP; ++p) kpf.p[p] = ld_row8_perm<DP>(smem, KI + p * IMG, kj, 32 * ks, q4);
                dvacc[t] = mfma_parts<SP, NP>(rtf, kpf, dvacc[t]);
            }
        }
        // the K row fragments are first needed by the score tiles: read here, not above (at D = 128 the state products hold
        // the whole register file; 16 more live registers there spilled into the chunk loop)
        if constexpr (DP == 128) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int p = 0; p < NP; ++p) kf[ks].p[p] = ld_row8<DP>(smem, KI + p * IMG, kj, 4 * ks + q4);
        Frag<2> tf[2], pf[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            f32x4 tt[2], pp[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int it = 2 * s + e;                            // 16 queries of the chunk
                f32x4 u = *reinterpret_cast<const f32x4*>(smem + ES + (16 * it + 4 * q4) * 4);    // e_i per register row
                f32x4 sc = {0, 0, 0, 0};
                if (it >= wk) {
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        Frag<NP> gf, qf;
#pragma unroll
                        for (int p = 0; p < NP; ++p) {
                            gf.p[p] = ld_row8<DP>(smem, GI + p * IMG, 16 * it + r, 4 * ks + q4);
                            qf.p[p] = ld_row8<DP>(smem, QI + p * IMG, 16 * it + r, 4 * ks + q4);
                        }
                        u = mfma_parts<NP, NP>(gf, vf[ks], u);          // ghat_i . v_j + e_i
                        sc = mfma_parts<NP, NP>(qf, kf[ks], sc);        // q_i . k_j
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bool keep = (it > wk) || (it == wk && (4 * q4 + i) >= r);
                    // rows past N carry w_i = 0 -> ghat = 0, e = 0, and P is multiplied by ghat rows (zero) below
                    tt[e][i] = keep ? a * u[i] : 0.f;
                    pp[e][i] = keep ? 1.0f + a * sc[i] : 0.f;
                }
            }
            bf16x4 h0, l0, h1, l1;
            split4(tt[0], h0, l0); split4(tt[1], h1, l1);
            tf[s].p[0] = cat4(h0, h1); tf[s].p[1] = cat4(l0, l1);
            split4(pp[0], h0, l0); split4(pp[1], h1, l1);
            pf[s].p[0] = cat4(h0, h1); pf[s].p[1] = cat4(l0, l1);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (2 * s + 1 >= wk) {
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    Frag<NP> qtf, gtf;
#pragma unroll
                    for (int p = 0; p < NP; ++p) {
                        qtf.p[p] = ld_tr8<DP>(smem, QI + p * IMG, 32 * s, 16 * (t0 + t), lane);
                        gtf.p[p] = ld_tr8<DP>(smem, GI + p * IMG, 32 * s, 16 * (t0 + t), lane);
                    }
                    dkacc[t] = mfma_parts<NP, 2>(qtf, tf[s], dkacc[t]);      // += Q[i][m] a T_ij
                    dvacc[t] = mfma_parts<NP, 2>(gtf, pf[s], dvacc[t]);      // += ghat[i][d] P_ij
                }
            }
        }
        // ---- phase B: R2[m = 16w + r][d] += sum_i Q[i][m] ghat[i][d] -------------------------------------
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            Frag<NP> qf[NSL];
#pragma unroll
            for (int sl = 0; sl < NSL; ++sl)
#pragma unroll
                for (int p = 0; p < NP; ++p) qf[sl].p[p] = ld_tr8<DP>(smem, QI + p * IMG, 32 * s, 16 * (wk + 4 * sl), lane);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                Frag<NP> gtf;
#pragma unroll
                for (int p = 0; p < NP; ++p) gtf.p[p] = ld_tr8<DP>(smem, GI + p * IMG, 32 * s, 16 * (t0 + dt), lane);
#pragma unroll
                for (int sl = 0; sl < NSL; ++sl) r2acc[sl][dt] = mfma_parts<NP, NP>(gtf, qf[sl], r2acc[sl][dt]);
            }
        }
        float krs = 0.f;
        if constexpr (FUSEK) {
            // prologue backward on the dK tile: row sums (for the mean over D) and the dot with the normalised key rows (still in
            // the K image until B2); lane (r, q4) holds columns 16 (t0 + t) + 4 q4 + i of key row kj
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                const int c0 = 16 * (t0 + t) + 4 * q4;
                f32x4 y = {0, 0, 0, 0};
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    const bf16x4 yb = *reinterpret_cast<const bf16x4*>(smem + KI + p * IMG + img_off<DP>(kj, c0 >> 3) + ((c0 & 7) << 1));
#pragma unroll
                    for (int i = 0; i < 4; ++i) y[i] += (float)yb[i];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    krs += dkacc[t][i];
                    kdot = fmaf(dkacc[t][i], y[i], kdot);
                }
            }
            if constexpr (FUSEK == 1) {
                krs += __shfl_xor(krs, 16, 64);
                krs += __shfl_xor(krs, 32, 64);
                if constexpr (NW == 8) {
                    if (q4 == 0) reinterpret_cast<float*>(smem + RSUM)[16 * w + r] = krs;
                }
            }
        }
        __syncthreads();                                             // B2
        if constexpr (FUSEK == 1) {
            if constexpr (NW == 8) krs += reinterpret_cast<const float*>(smem + RSUM)[16 * (w ^ 4) + r];     // the other column half
            const float kmean = krs * invD, kinv = prm.kscale[bh];     // read here: one more live value across the loop spills at D = 128
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) dkacc[t][i] = (dkacc[t][i] - kmean) * kinv;      // padded columns are not stored
        }
        if constexpr (NW == 4) {
            store_tile16<DP>(smem + w * (16 * DP * 4), dkacc, 1.0f, lane, prm.dk, prm.grad_dtype,   // a is already folded in
                             ((int64_t)bh * N + n0 + 16 * w) * D, n0 + 16 * w, N, D);
            store_tile16<DP>(smem + DP * 256 + w * (16 * DP * 4), dvacc, 1.0f, lane, prm.dv, prm.grad_dtype,
                             ((int64_t)bh * N + n0 + 16 * w) * D, n0 + 16 * w, N, D);
        } else {
            // the Q / K / V / G images are free after B2: 8 waves x 2 areas of 16 x (DP/2) x 4 bytes
            constexpr int AREA = 16 * 16 * MT * 4;
            store_cols16<16 * MT>(smem + (2 * w) * AREA, dkacc, 1.0f, lane, prm.dk, prm.grad_dtype,
                                  ((int64_t)bh * N + n0 + 16 * wk) * D, 16 * t0, n0 + 16 * wk, N, D);
            store_cols16<16 * DT>(smem + (2 * w + 1) * AREA, dvacc, 1.0f, lane, prm.dv, prm.grad_dtype,
                                  ((int64_t)bh * N + n0 + 16 * wk) * D, 16 * t0, n0 + 16 * wk, N, D);
        }
        if (c > c_begin) {
#pragma unroll
            for (int sl = 0; sl < NSL; ++sl) publish_state<DP, SP, DT>(smem, R2I, SIMG, r2acc[sl], a, 16 * (wk + 4 * sl) + r, q4, t0);
        }
        __syncthreads();
    }
    if constexpr (FUSEK) {
        // one record per block: T = sum_n dk'_n . y_n (= sum_n dq'_n . y^q_n = sum_ij dS_ij s_ij), for the one-row fix-ups
        kdot = wave_sum(kdot);
        float* fd = reinterpret_cast<float*>(smem + RSUM);
        if (lane == 0) fd[w] = kdot;
        __syncthreads();
        if (tid == 0) {
            float sd = 0.f;
            for (int i = 0; i < NW; ++i) sd += fd[i];
            prm.kpart_dot[(int64_t)bh * prm.nseg + seg] = sd;          // T = sum dS . s: the same number for the q side
        }
    }
}

template <int DP, typename TIN, bool NORM, bool BUF, int FUSEK, bool FUSEQ = false>
static int launch_lin_bwd_b(const LinBwdParams& prm, int BH, hipStream_t stream, const fastmax_problem& prob) {
    constexpr int NP = InTraits<TIN>::NP, EPL = InTraits<TIN>::EPL;
    // four waves where two workgroups fit a CU (bf16 D <= 64: both kernels; two-part D <= 64: dQ only), eight waves
    // (one workgroup per CU, column halves per wave) everywhere else
    constexpr int NWQ = DP == 64 ? 4 : 8, NWKV = (DP == 64 && NP == 1) ? 4 : 8;
    constexpr int RPPQ = 64 * NWQ / (DP / EPL), RPPKV = 64 * NWKV / (DP / EPL);
    constexpr int IMG = 64 * DP * 2, SIMG = DP * DP * 2;
    constexpr int lds_q = 3 * NP * IMG + NP * SIMG + 2 * DP * 4 + RPPQ * DP * 4 + 512 + (NWQ == 8 ? 8 * 16 * (DP / 2) * 4 : 0) + 512;
    constexpr int lds_kv = 4 * NP * IMG + NP * SIMG + 4 * DP * 4 + 2 * RPPKV * DP * 4 + 256 + 640;
    static_assert(lds_kv <= 160 * 1024 && lds_q <= 160 * 1024, "LDS budget");
    static_assert(NWKV == 4 || 8 * 2 * 16 * (DP / 2) * 4 <= 4 * NP * IMG, "dK/dV staging areas fit the freed images");
    auto kq = bwd_p1_dq_kernel<DP, TIN, NWQ, NORM, BUF, FUSEQ>;
    auto kkv = bwd_p1_dkv_kernel<DP, TIN, NWKV, NORM, BUF, FUSEK>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kq), hipFuncAttributeMaxDynamicSharedMemorySize, lds_q);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(kkv), hipFuncAttributeMaxDynamicSharedMemorySize, lds_kv);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(kq, dim3(BH * prm.nseg), dim3(64 * NWQ), lds_q, stream, prm);
    if (prm.nseg > 1) {
        const int rc = launch_split_rstates(prm.q, prm.qs, prm.go, prm.gos, prm.g, prm.c, const_cast<float*>(prm.rstate), prob,
                                            SplitPlan{prm.nseg, prm.cps}, DP, stream, prm.qscale);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(kkv, dim3(BH * prm.nseg), dim3(64 * NWKV), lds_kv, stream, prm);
    if constexpr (FUSEK != 0) {
        int rc = 0;
        if constexpr (FUSEK == 1)
            rc = launch_normalize_fixadd(prm.k, prm.ks, prob.in_dtype, prm.kscale, prm.kpart_dot, prm.k_nstar, prm.nseg, prob.B, prob.H,
                                         prm.N, prm.D, prm.dk, stream);
        if constexpr (FUSEQ) {
            if (rc) return rc;
            rc = launch_normalize_fixadd(prm.q, prm.qs, prob.in_dtype, prm.qscale, prm.kpart_dot, prm.q_nstar, prm.nseg, prob.B, prob.H,
                                         prm.N, prm.D, prm.dq, stream);
        }
        return rc;
    }
    return (int)hipGetLastError();
}

template <int DP, typename TIN, bool NORM>
static int launch_lin_bwd_t(const LinBwdParams& prm, int BH, hipStream_t stream, const fastmax_problem& prob) {
    // buffer-descriptor tile loads when every (b,h) slab spans less than their 31-bit offsets (all but pathological strides)
    const int es = (int)sizeof(TIN);
    const bool buf = quad32_span_ok(prm.qs.sn, prm.N, prm.D, es) && quad32_span_ok(prm.ks.sn, prm.N, prm.D, es) &&
                     quad32_span_ok(prm.vs.sn, prm.N, prm.D, es) && quad32_span_ok(prm.gos.sn, prm.N, prm.D, es);
    if constexpr (NORM) {
        if (prm.kpart_dot && prm.q_nstar && prm.k_nstar)
            return buf ? launch_lin_bwd_b<DP, TIN, true, true, 1, true>(prm, BH, stream, prob) : launch_lin_bwd_b<DP, TIN, true, false, 1, true>(prm, BH, stream, prob);
        if (prm.kpart_dot && prm.q_nstar)          // q side only (grouped-query heads)
            return buf ? launch_lin_bwd_b<DP, TIN, true, true, 2, true>(prm, BH, stream, prob) : launch_lin_bwd_b<DP, TIN, true, false, 2, true>(prm, BH, stream, prob);
        if (prm.kpart_dot) return buf ? launch_lin_bwd_b<DP, TIN, true, true, 1>(prm, BH, stream, prob) : launch_lin_bwd_b<DP, TIN, true, false, 1>(prm, BH, stream, prob);
    }
    return buf ? launch_lin_bwd_b<DP, TIN, NORM, true, 0>(prm, BH, stream, prob) : launch_lin_bwd_b<DP, TIN, NORM, false, 0>(prm, BH, stream, prob);
}

bool lin_bwd_supported(const fastmax_problem& p);
static size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }
// per (head, segment) records of the fused prologue backward: one float (32 segments at most)
static size_t lin_bwd_fuse_bytes(const fastmax_problem& p) { return align16((size_t)p.B * p.H * 32 * 4); }
// workspace = [ c (B,H,N) | forward-scan states | reverse-scan states | fused-prologue records ]   (the states only when the sequence is split)
size_t lin_bwd_workspace(const fastmax_problem& p) {
    size_t bytes = align16(sizeof(float) * (size_t)p.B * p.H * p.Nq);
    if (lin_bwd_supported(p)) bytes += 2 * align16(split_workspace_bytes(p, p.D <= 64 ? 64 : 128)) + lin_bwd_fuse_bytes(p);
    return bytes;
}

bool lin_bwd_supported(const fastmax_problem& p) {
    // D <= 64 for every dtype; 64 < D <= 128 for bf16 (one operand part: the images and the D x D state fit the 160 KB LDS)
    if (!(p.p == 1 && p.causal) || p.D > 128 || (p.D > 64 && p.in_dtype != FASTMAX_BF16)) return false;
    const int epl = p.in_dtype == FASTMAX_F32 ? 4 : 8;
    // linear time pays off once the O(N^2) tiles outgrow the carried-state work
    return (p.D % epl) == 0 && p.Nq >= 512;
}

int launch_bwd_lin(const BwdArgs& a) {
    if (!lin_bwd_supported(a.prob)) return FASTMAX_E_BAD_SHAPE;
    if (a.workspace_bytes < lin_bwd_workspace(a.prob) || !a.workspace) return FASTMAX_E_WORKSPACE;
    const int dp = a.prob.D <= 64 ? 64 : 128;
    const SplitPlan plan = split_plan(a.prob);
    char* ws = reinterpret_cast<char*>(a.workspace);
    float* cbuf = reinterpret_cast<float*>(ws);
    const size_t coff = align16(sizeof(float) * (size_t)a.prob.B * a.prob.H * a.prob.Nq), sbytes = align16(split_workspace_bytes(a.prob, dp));
    // forward-scan states: the forward's own records when the caller kept its workspace, else recomputed below
    const float* fstate = a.fwd_states ? a.fwd_states : reinterpret_cast<const float*>(ws + coff);
    float* rstate = reinterpret_cast<float*>(ws + coff + sbytes);
    LinBwdParams prm{a.q, a.k, a.v, a.o, a.grad_o, a.g, a.qs, a.ks, a.vs, a.gos, a.dq, a.dk, a.dv,
                     cbuf, a.prob.H, a.prob.Nq, a.prob.D, a.prob.in_dtype, a.prob.out_dtype, a.prob.a,
                     fstate, rstate, plan.nseg, plan.cps, a.qscale, a.kscale, nullptr, nullptr, nullptr};
    if (a.fuse_prologue && a.qscale && a.kscale) {
        prm.kpart_dot = reinterpret_cast<float*>(ws + coff + 2 * sbytes);
        if (a.fuse_prologue & 1) prm.k_nstar = a.k_nstar;
        if (a.fuse_prologue & 2) prm.q_nstar = a.q_nstar;
    }
    const int BH = a.prob.B * a.prob.H;
    if (plan.nseg > 1 && !a.fwd_states) {
        // forward-scan states (sum k v^T, sum k) exactly as the forward's; the reverse-scan states need c_i, which the dQ
        // kernel writes, so they are computed between the two main kernels
        FwdArgs fa{a.prob, a.q, a.k, a.v, a.qs, a.ks, a.vs, nullptr, nullptr, ws + coff, sbytes, a.stream};
        const int rc = launch_split_states(fa, plan, dp, a.kscale);          // kscale: K is raw (linearmax training route)
        if (rc) return rc;
    }
    if (a.qscale && a.kscale) {
        switch (a.prob.in_dtype) {
            case FASTMAX_F32: return launch_lin_bwd_t<64, float, true>(prm, BH, a.stream, a.prob);
            case FASTMAX_BF16: return a.prob.D <= 64 ? launch_lin_bwd_t<64, bf16_t, true>(prm, BH, a.stream, a.prob) : launch_lin_bwd_t<128, bf16_t, true>(prm, BH, a.stream, a.prob);
            case FASTMAX_F16: return launch_lin_bwd_t<64, f16_t, true>(prm, BH, a.stream, a.prob);
        }
        return FASTMAX_E_BAD_DTYPE;
    }
    switch (a.prob.in_dtype) {
        case FASTMAX_F32: return launch_lin_bwd_t<64, float, false>(prm, BH, a.stream, a.prob);
        case FASTMAX_BF16: return a.prob.D <= 64 ? launch_lin_bwd_t<64, bf16_t, false>(prm, BH, a.stream, a.prob) : launch_lin_bwd_t<128, bf16_t, false>(prm, BH, a.stream, a.prob);
        case FASTMAX_F16: return launch_lin_bwd_t<64, f16_t, false>(prm, BH, a.stream, a.prob);
    }
    return FASTMAX_E_BAD_DTYPE;
}

}  // namespace fastmax
