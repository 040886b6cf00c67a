// Sequence split for the linear-time fastmax kernels: when B*H workgroups cannot fill 256 CUs the
// sequence of every head is cut into `nseg` segments that run as independent workgroups.
//   1. p1_state_kernel     per (head, segment < nseg-1): local sums  S2 = sum k v^T, S1 = sum v, ksum = sum k
//                          (the same K^T V MFMA step and exact fp32 column sums as the main kernel)
//   2. p1_state_prefix     inclusive prefix of the records over the segments of a head (the causal
//                          cumulative sum at segment granularity)
//   3. the main kernel starts segment s from record s-1.
// Cost: K and V of all but the last segment are read twice (+ <= 50 % of the K,V bytes, 0 when nseg = 1).
#include "fastmax_mfma_common.h"

#include <cstdlib>

namespace fastmax {

struct StateParams {
    const void *k, *v;
    Strides3 ks, vs;
    float* state;
    const float* kscale;      // fused linearmax: K rows are (k - mean) * kscale[bh]
    int H, N, D, nseg, cps;
    const float *g, *c;       // RS (reverse states for the backward): k = q, v = grad_o, rows of v scaled by w_i = 1/g_i,
                              // "ksum" weighted by e_i = -w_i c_i
    // NORM = 2 (the linearmax statistics ride on this pass): K rows are centred but NOT yet scaled (the state is linear in K, so
    // the prefix pass applies the scale), every block leaves max ||row - mean||^2 of its rows in partials[which][bh][seg]
    const void* q;
    Strides3 qs;
    unsigned long long* partials;   // [2 (q, k)][B*H][pw] keys: (bits of max ||row - mean||^2) << 32 | ~row
    int BH, pw;
    int rs_first = 1;         // RS: the first segment a block computes (1: the causal scans never need segment 0's own sums;
                              // 0: the unmasked backward wants the total)
};
constexpr int STAT_LT = 4;    // chunks (of 64 rows) per statistics-only block

// RS = false: forward states of segments 0 .. nseg-2 (record seg).  RS = true: the reverse-scan states of the p=1 backward
// (fastmax_mfma_bwd_lin.hip) of segments 1 .. nseg-1 (record seg-1): R2 = sum q ghat^T, R1 = sum ghat, rq = sum q e.
// block = 4 DP threads: one wave per 16-column slab of the state (DP / 16 waves; round 3: eight waves at D = 128 instead of four
// with two slabs each -- the pass is latency-bound, one workgroup per CU)
// NORM = 2: a 1-D grid.  The first B*H*(nseg-1) blocks are the state blocks above (they run the longest, so they are dispatched
// first) and also leave the statistic of their K rows; the blocks after them are statistics-only (no images, no barriers, no
// MFMA): STAT_LT chunks each, all loads issued at once -- per head ceil(nchunks / STAT_LT) blocks for Q, then the blocks that cover
// K's last segment.  Word j of a head: Q block j | K: segment (j < nseg-1) or last-segment block j - (nseg-1).
template <int DP, typename TIN, int NORM, bool RS = false>
__global__ __launch_bounds__(4 * DP, 2) void p1_state_kernel(StateParams prm) {
    constexpr int NP = InTraits<TIN>::NP, EPL = InTraits<TIN>::EPL;
    constexpr int NT = 4 * DP;
    constexpr int C = 64, IMG = C * DP * 2;
    constexpr int KI = 0, VI = NP * IMG, PARTV = 2 * NP * IMG;
    constexpr int COLS = DP / EPL, RPP = NT / COLS, NPASS = C / RPP;
    constexpr int PARTK = PARTV + RPP * DP * 4;
    constexpr int MT = DP / 16, NSL = 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q4 = lane >> 4;
    const int nchunks_all = (prm.N + 63) / 64;
    int bh = blockIdx.y, seg = (int)blockIdx.x + (RS ? prm.rs_first : 0);
    bool statq = false, maxonly = false;                                    // block-uniform
    int light_c0 = 0, word = 0;
    if constexpr (NORM == 2) {
        const int nheavy = prm.BH * (prm.nseg - 1), idx = blockIdx.x;
        if (idx < nheavy) {
            bh = idx / (prm.nseg - 1);
            seg = idx - bh * (prm.nseg - 1);
            word = seg;
        } else {
            const int last_c0 = (prm.nseg - 1) * prm.cps;
            const int nlq = (nchunks_all + STAT_LT - 1) / STAT_LT, nlk = (nchunks_all - last_c0 + STAT_LT - 1) / STAT_LT;
            const int li = idx - nheavy;
            bh = li / (nlq + nlk);
            const int j = li - bh * (nlq + nlk);
            maxonly = true;
            statq = j < nlq;
            light_c0 = statq ? j * STAT_LT : last_c0 + (j - nlq) * STAT_LT;
            word = statq ? j : prm.nseg - 1 + (j - nlq);
        }
    }
    const int b = bh / prm.H, h = bh % prm.H;
    const int N = prm.N, D = prm.D;
    const Strides3 kst = statq ? prm.qs : prm.ks;
    const TIN* kb = reinterpret_cast<const TIN*>(statq ? prm.q : prm.k) + (int64_t)b * kst.sb + (int64_t)h * kst.sh;
    const TIN* vb = reinterpret_cast<const TIN*>(prm.v) + (int64_t)b * prm.vs.sb + (int64_t)h * prm.vs.sh;
    const int srow = tid / COLS, scol = tid % COLS;
    const bool colok = scol * EPL < D;
    float ksc = 1.f;
    if constexpr (NORM == 1) ksc = prm.kscale[bh];
    const float ksc_c = colok ? ksc : 0.f;
    const float invD = 1.0f / (float)D;
    const int nchunks = (N + C - 1) / C;
    const int c_begin = seg * prm.cps, c_end = min(nchunks, c_begin + prm.cps);

    u32x4 rk[NPASS], rv[NPASS];
    const TileLoader<TIN, NPASS, RPP> kload(kb, kst.sn, N, D, DP, srow, scol), vload(vb, prm.vs.sn, N, D, DP, srow, scol);
    float rgg[NPASS], rcc[NPASS];                                  // RS: g_i, c_i of the rows in flight (raw: see bwd_p1_dq_kernel)
    auto issue = [&](int n0) {
        kload.load(n0 / C, rk);
        vload.load(n0 / C, rv);
        if constexpr (RS) {
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps) {
                const int gi = n0 + srow + ps * RPP, gc = gi < N ? gi : N - 1;
                rgg[ps] = prm.g[(int64_t)bh * N + gc];
                rcc[ps] = prm.c[(int64_t)bh * N + gc];
            }
        }
    };
    unsigned long long best = 0ull;                                // (squared norm bits, ~row): max = largest norm, lowest row on ties
    auto keep_best = [&](float nn, int row) __attribute__((always_inline)) {
        const unsigned long long key = ((unsigned long long)__float_as_uint(nn) << 32) | (unsigned long long)(0xffffffffu - (unsigned)row);
        best = key > best ? key : best;
    };
    f32x4 s2acc[NSL][MT];
#pragma unroll
    for (int sl = 0; sl < NSL; ++sl)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) s2acc[sl][mt] = f32x4{0, 0, 0, 0};
    float ck[EPL], cv[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) { ck[e] = 0.f; cv[e] = 0.f; }

    auto rowstat = [&](const u32x4& piece, int row) __attribute__((always_inline)) {
        float xk[EPL];
        piece_to_float<TIN>(piece, xk);
        float sk = 0.f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) sk += xk[e];
        const float nmk = -rowsum_all<COLS>(sk) * invD * ksc_c;
        float nn = 0.f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const float xc = fmaf(xk[e], ksc_c, nmk);
            nn = fmaf(xc, xc, nn);
        }
        keep_best(rowsum_all<COLS>(nn), row);
    };
    if constexpr (NORM == 2) {
        if (maxonly) {
            u32x4 t[STAT_LT][NPASS];
#pragma unroll
            for (int i = 0; i < STAT_LT; ++i) kload.load(light_c0 + i, t[i]);          // tiles past the tensor load as zeros
#pragma unroll
            for (int i = 0; i < STAT_LT; ++i)
#pragma unroll
                for (int ps = 0; ps < NPASS; ++ps) rowstat(t[i][ps], (light_c0 + i) * C + srow + ps * RPP);
        }
    }
    if (!maxonly) issue(c_begin * C);
    for (int c = maxonly ? c_end : c_begin; c < c_end; ++c) {
        const int n0 = c * C;
        __syncthreads();                                           // previous chunk's images consumed
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int row = srow + ps * RPP;
            float xk[EPL], xv[EPL];
            piece_to_float<TIN>(rk[ps], xk);
            piece_to_float<TIN>(rv[ps], xv);
            if constexpr (RS) {
                const int gi = n0 + row;
                const float wi = gi < N ? 1.0f / rgg[ps] : 0.f;
                const float ei = -wi * rcc[ps];
#pragma unroll
                for (int e = 0; e < EPL; ++e) xv[e] *= wi;
                if constexpr (NORM == 1) {                               // linearmax training route: q arrives raw
                    normalize_piece<COLS, EPL>(xk, ksc_c, invD);
                    stage_floats<DP, EPL, NP>(smem, KI, row, scol, xk);
                } else {
                    stage_piece<DP, TIN>(smem, KI, row, scol, rk[ps]);
                }
                stage_floats<DP, EPL, NP>(smem, VI, row, scol, xv);
#pragma unroll
                for (int e = 0; e < EPL; ++e) { ck[e] = fmaf(xk[e], ei, ck[e]); cv[e] += xv[e]; }
                continue;
            }
            if constexpr (NORM) {
                float sk = 0.f;
#pragma unroll
                for (int e = 0; e < EPL; ++e) sk += xk[e];
                // one fma per element; rows past N were loaded as zeros (mean 0), padded columns have a zero scale
                const float nmk = -rowsum_all<COLS>(sk) * invD * ksc_c;
#pragma unroll
                for (int e = 0; e < EPL; ++e) xk[e] = fmaf(xk[e], ksc_c, nmk);
                if constexpr (NORM == 2) {
                    float nn = 0.f;
#pragma unroll
                    for (int e = 0; e < EPL; ++e) nn = fmaf(xk[e], xk[e], nn);
                    keep_best(rowsum_all<COLS>(nn), n0 + row);
                }
                stage_floats<DP, EPL, NP>(smem, KI, row, scol, xk);
            } else {
                stage_piece<DP, TIN>(smem, KI, row, scol, rk[ps]);
            }
            stage_piece<DP, TIN>(smem, VI, row, scol, rv[ps]);
#pragma unroll
            for (int e = 0; e < EPL; ++e) { ck[e] += xk[e]; cv[e] += xv[e]; }
        }
        if (c + 1 < c_end) issue(n0 + C);
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int sl = 0; sl < NSL; ++sl) {
                Frag<NP> vf;
#pragma unroll
                for (int p = 0; p < NP; ++p) vf.p[p] = ld_tr8<DP>(smem, VI + p * IMG, 32 * s, 16 * (w + 4 * sl), lane);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    Frag<NP> kf;
#pragma unroll
                    for (int p = 0; p < NP; ++p) kf.p[p] = ld_tr8<DP>(smem, KI + p * IMG, 32 * s, 16 * mt, lane);
                    s2acc[sl][mt] = mfma_parts<NP, NP>(kf, vf, s2acc[sl][mt]);
                }
            }
        }
    }
    if constexpr (NORM == 2) {
        __shared__ unsigned long long wmax[NT / 64];
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned long long o = __shfl_xor(best, off, 64);
            best = o > best ? o : best;
        }
        if (lane == 0) wmax[w] = best;
        __syncthreads();
        if (tid == 0) {
            unsigned long long m = wmax[0];
#pragma unroll
            for (int i = 1; i < NT / 64; ++i) m = wmax[i] > m ? wmax[i] : m;
            prm.partials[((int64_t)(statq ? 0 : 1) * prm.BH + bh) * prm.pw + word] = m;
        }
        if (maxonly) return;
    }
    // record = [S2 (DP x DP, row-major [m][d]) | S1 (DP) | ksum (DP)]
    float* rec = prm.state + ((int64_t)bh * (prm.nseg - 1) + seg - (RS ? prm.rs_first : 0)) * (DP * DP + 2 * DP);
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        reinterpret_cast<float*>(smem + PARTK)[srow * DP + scol * EPL + e] = ck[e];
        reinterpret_cast<float*>(smem + PARTV)[srow * DP + scol * EPL + e] = cv[e];
    }
#pragma unroll
    for (int sl = 0; sl < NSL; ++sl)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int i = 0; i < 4; ++i) rec[(16 * mt + 4 * q4 + i) * DP + 16 * (w + 4 * sl) + r] = s2acc[sl][mt][i];
    __syncthreads();
    for (int t = tid; t < 2 * DP; t += NT) {
        const int col = t % DP;
        const float* part = reinterpret_cast<const float*>(smem + (t < DP ? PARTV : PARTK));
        float s = 0.f;
        for (int g16 = 0; g16 < RPP; ++g16) s += part[g16 * DP + col];
        rec[DP * DP + (t < DP ? 0 : DP) + col] = s;
    }
}

// inclusive prefix (reverse = 1: suffix) over the nseg-1 records of one head; grid = (ceil(REC/256), B*H)
__global__ __launch_bounds__(256) void p1_state_prefix_kernel(float* state, int nrec, int rec_floats, int reverse) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= rec_floats) return;
    float* base = state + (int64_t)blockIdx.y * nrec * rec_floats + e;
    float acc = 0.f;
    for (int i = 0; i < nrec; ++i) {
        const int s = reverse ? nrec - 1 - i : i;
        acc += base[(int64_t)s * rec_floats];
        base[(int64_t)s * rec_floats] = acc;
    }
}

// NORM = 2 companion: folds the statistic words into inv_q, inv_k = 1 / sqrt(max) (block 0 of a head writes them for the main
// kernel) and runs the inclusive prefix with the K scale applied to what is linear in K: S2 and ksum (S1 = sum v is not)
__global__ __launch_bounds__(256) void p1_state_prefix_scale_kernel(float* state, int nrec, int rec_floats, int dp,
                                                                    const unsigned long long* partials, int pw, int nq, int nk,
                                                                    float* inv_q, float* inv_k, int* nstar_q, int* nstar_k) {
    __shared__ unsigned long long red[8];
    const int bh = blockIdx.y, BH = gridDim.y, tid = threadIdx.x;
    // every block of a head folds K's keys (a few dozen); block 0 also Q's and writes both for the main kernel
    unsigned long long mk = 0ull, mq = 0ull;
    for (int j = tid; j < nk; j += 256) { const unsigned long long o = partials[((int64_t)BH + bh) * pw + j]; mk = o > mk ? o : mk; }
    if (blockIdx.x == 0)
        for (int j = tid; j < nq; j += 256) { const unsigned long long o = partials[(int64_t)bh * pw + j]; mq = o > mq ? o : mq; }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long ok = __shfl_xor(mk, off, 64), oq = __shfl_xor(mq, off, 64);
        mk = ok > mk ? ok : mk;
        mq = oq > mq ? oq : mq;
    }
    if ((tid & 63) == 0) { red[tid >> 6] = mk; red[4 + (tid >> 6)] = mq; }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) { mk = red[i] > mk ? red[i] : mk; mq = red[4 + i] > mq ? red[4 + i] : mq; }
    const float ksc = 1.0f / sqrtf(__uint_as_float((unsigned)(mk >> 32)));          // squared norms: non-negative floats
    if (blockIdx.x == 0 && tid == 0) {
        inv_q[bh] = 1.0f / sqrtf(__uint_as_float((unsigned)(mq >> 32)));
        inv_k[bh] = ksc;
        if (nstar_q) nstar_q[bh] = (int)(0xffffffffu - (unsigned)(mq & 0xffffffffull));
        if (nstar_k) nstar_k[bh] = (int)(0xffffffffu - (unsigned)(mk & 0xffffffffull));
    }
    const int e = blockIdx.x * 256 + tid;
    if (e >= rec_floats) return;
    const float sc = (e >= dp * dp && e < dp * dp + dp) ? 1.f : ksc;
    float* base = state + (int64_t)bh * nrec * rec_floats + e;
    float acc = 0.f;
    for (int i = 0; i < nrec; ++i) {
        acc = fmaf(base[(int64_t)i * rec_floats], sc, acc);
        base[(int64_t)i * rec_floats] = acc;
    }
}

SplitPlan split_plan(const fastmax_problem& p) {
    // aim at two workgroups per CU for D <= 64 and one for D > 64 (those kernels hold a 128 x 128 state: one 8-wave
    // workgroup per CU); FASTMAX_SPLIT_TARGET overrides for experiments (measured: 512 is the optimum at D = 64)
    static const int forced = [] { const char* e = getenv("FASTMAX_SPLIT_TARGET"); return e ? atoi(e) : 0; }();
    const int target = forced ? forced : (p.D > 64 ? 256 : 512);
    const int BH = p.B * p.H, nchunks = (p.Nq + 63) / 64;
    if (BH >= target * 3 / 4 || nchunks < 8) return SplitPlan{1, nchunks};
    int nseg = (target + BH - 1) / BH;
    if (nseg > nchunks / 4) nseg = nchunks / 4;                  // at least 4 chunks per segment
    if (nseg > 32) nseg = 32;
    if (nseg < 2) return SplitPlan{1, nchunks};
    const int cps = (nchunks + nseg - 1) / nseg;
    nseg = (nchunks + cps - 1) / cps;                            // drop empty trailing segments
    return SplitPlan{nseg, cps};
}

size_t split_workspace_bytes(const fastmax_problem& p, int dp) {
    const SplitPlan plan = split_plan(p);
    if (plan.nseg <= 1) return 0;
    return sizeof(float) * (size_t)p.B * p.H * (plan.nseg - 1) * ((size_t)dp * dp + 2 * dp);
}

template <int DP, typename TIN, int NORM, bool RS = false>
static int launch_state_t(const StateParams& prm, int BH, hipStream_t stream, float* inv_q = nullptr, float* inv_k = nullptr,
                          int* nstar_q = nullptr, int* nstar_k = nullptr) {
    constexpr int NP = InTraits<TIN>::NP, EPL = InTraits<TIN>::EPL, RPP = 4 * DP / (DP / EPL);
    constexpr int lds = 2 * NP * 64 * DP * 2 + 2 * RPP * DP * 4;
    auto kern = p1_state_kernel<DP, TIN, NORM, RS>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const int rec = DP * DP + 2 * DP;
    if constexpr (NORM == 2) {
        const int nchunks = (prm.N + 63) / 64, last_c0 = (prm.nseg - 1) * prm.cps;
        const int nlq = (nchunks + STAT_LT - 1) / STAT_LT, nlk = (nchunks - last_c0 + STAT_LT - 1) / STAT_LT;
        hipLaunchKernelGGL(kern, dim3(BH * (prm.nseg - 1 + nlq + nlk)), dim3(4 * DP), lds, stream, prm);
        hipLaunchKernelGGL(p1_state_prefix_scale_kernel, dim3((rec + 255) / 256, BH), dim3(256), 0, stream, prm.state, prm.nseg - 1,
                           rec, DP, prm.partials, prm.pw, nlq, prm.nseg - 1 + nlk, inv_q, inv_k, nstar_q, nstar_k);
        return (int)hipGetLastError();
    }
    hipLaunchKernelGGL(kern, dim3(prm.nseg - 1, BH), dim3(4 * DP), lds, stream, prm);
    // one record per head (two segments) is its own prefix
    if (prm.nseg > 2)
        hipLaunchKernelGGL(p1_state_prefix_kernel, dim3((rec + 255) / 256, BH), dim3(256), 0, stream, prm.state, prm.nseg - 1, rec, RS ? 1 : 0);
    return (int)hipGetLastError();
}
template <typename TIN, int NORM>
static int launch_state_d(const StateParams& prm, int BH, int dp, hipStream_t stream, float* inv_q = nullptr, float* inv_k = nullptr,
                          int* nstar_q = nullptr, int* nstar_k = nullptr) {
    if (dp == 64) return launch_state_t<64, TIN, NORM>(prm, BH, stream, inv_q, inv_k, nstar_q, nstar_k);
    return launch_state_t<128, TIN, NORM>(prm, BH, stream, inv_q, inv_k, nstar_q, nstar_k);
}

// reverse-scan states of the linear-time backward: q in the K role, grad_o (scaled by 1/g) in the V role
int launch_split_rstates(const void* q, Strides3 qs, const void* go, Strides3 gos, const float* g, const float* c, float* state,
                         const fastmax_problem& p, const SplitPlan& plan, int dp, hipStream_t stream, const float* qscale, int first_seg) {
    StateParams prm{q, go, qs, gos, state, qscale, p.H, p.Nq, p.D, plan.nseg, plan.cps, g, c, nullptr, Strides3{}, nullptr, 0, 0, first_seg};
    const int BH = p.B * p.H;
    if (qscale) {
        switch (p.in_dtype) {
            case FASTMAX_F32: return dp == 64 ? launch_state_t<64, float, 1, true>(prm, BH, stream) : FASTMAX_E_BAD_SHAPE;
            case FASTMAX_BF16: return dp == 64 ? launch_state_t<64, bf16_t, 1, true>(prm, BH, stream) : launch_state_t<128, bf16_t, 1, true>(prm, BH, stream);
            case FASTMAX_F16: return dp == 64 ? launch_state_t<64, f16_t, 1, true>(prm, BH, stream) : FASTMAX_E_BAD_SHAPE;
        }
        return FASTMAX_E_BAD_DTYPE;
    }
    switch (p.in_dtype) {
        case FASTMAX_F32: return dp == 64 ? launch_state_t<64, float, 0, true>(prm, BH, stream) : FASTMAX_E_BAD_SHAPE;
        case FASTMAX_BF16: return dp == 64 ? launch_state_t<64, bf16_t, 0, true>(prm, BH, stream) : launch_state_t<128, bf16_t, 0, true>(prm, BH, stream);
        case FASTMAX_F16: return dp == 64 ? launch_state_t<64, f16_t, 0, true>(prm, BH, stream) : FASTMAX_E_BAD_SHAPE;
    }
    return FASTMAX_E_BAD_DTYPE;
}

int launch_split_states(const FwdArgs& a, const SplitPlan& plan, int dp, const float* kscale) {
    StateParams prm{a.k, a.v, a.ks, a.vs, reinterpret_cast<float*>(a.workspace), kscale, a.prob.H, a.prob.Nq, a.prob.D,
                    plan.nseg, plan.cps, nullptr, nullptr, nullptr, Strides3{}, nullptr, 0, 0};
    const int BH = a.prob.B * a.prob.H;
    const bool norm = kscale != nullptr;
    switch (a.prob.in_dtype) {
        case FASTMAX_F32: return norm ? launch_state_d<float, 1>(prm, BH, dp, a.stream) : launch_state_d<float, 0>(prm, BH, dp, a.stream);
        case FASTMAX_BF16: return norm ? launch_state_d<bf16_t, 1>(prm, BH, dp, a.stream) : launch_state_d<bf16_t, 0>(prm, BH, dp, a.stream);
        case FASTMAX_F16: return norm ? launch_state_d<f16_t, 1>(prm, BH, dp, a.stream) : launch_state_d<f16_t, 0>(prm, BH, dp, a.stream);
    }
    return FASTMAX_E_BAD_DTYPE;
}

// linearmax with the statistics not computed yet (FwdArgs::stats): one launch leaves the segment states of the UNSCALED centred K
// and the statistic words of Q and K, the prefix pass turns the words into inv_q / inv_k and scales the records
int launch_split_states_stats(const FwdArgs& a, const SplitPlan& plan, int dp) {
    const LinearmaxStats& st = *a.stats;
    StateParams prm{a.k, a.v, a.ks, a.vs, reinterpret_cast<float*>(a.workspace), nullptr, a.prob.H, a.prob.Nq, a.prob.D,
                    plan.nseg, plan.cps, nullptr, nullptr, a.q, a.qs, reinterpret_cast<unsigned long long*>(st.partials), a.prob.B * a.prob.H,
                    (a.prob.Nq + 255) / 256 + 32};
    const int BH = a.prob.B * a.prob.H;
    if (BH > 65535) return FASTMAX_E_BAD_SHAPE;
    switch (a.prob.in_dtype) {
        case FASTMAX_F32: return launch_state_d<float, 2>(prm, BH, dp, a.stream, st.inv_q, st.inv_k, st.nstar_q, st.nstar_k);
        case FASTMAX_BF16: return launch_state_d<bf16_t, 2>(prm, BH, dp, a.stream, st.inv_q, st.inv_k, st.nstar_q, st.nstar_k);
        case FASTMAX_F16: return launch_state_d<f16_t, 2>(prm, BH, dp, a.stream, st.inv_q, st.inv_k, st.nstar_q, st.nstar_k);
    }
    return FASTMAX_E_BAD_DTYPE;
}

// statistics for a linearmax forward whose launcher found them missing: with the sequence split they ride on the state pass
// (above), without it they are the paired statistics pass.  Returns 0 and leaves inv_q / inv_k valid for the launches after it.
int linearmax_stats_and_states(const FwdArgs& a, const SplitPlan& plan, int dp) {
    const LinearmaxStats& st = *a.stats;
    if (plan.nseg > 1) return launch_split_states_stats(a, plan, dp);
    return launch_normalize_stats2(a.q, a.qs, a.k, a.ks, a.prob.in_dtype, st.inv_q, st.inv_k, a.prob.B, a.prob.H, a.prob.Nq, a.prob.D,
                                   st.partials, a.stream, st.nstar_q, st.nstar_k);
}

// ------------------------------------------------------------------------------------------------------------------
// Unmasked first order in linear time (fastmax.py:258-271 / fastmax_hack.py:6-33 without the (N,D,D) temporaries):
//     o_i = (S1 + a S2^T q_i) / (g0 + a q_i . ksum)      with the TOTAL sums S2 = sum_j k_j v_j^T, S1 = sum_j v_j, ksum = sum_j k_j
// = the state pass above over all of K, V (per-segment records + inclusive prefix: the last record is the total) and one
// D x D product per query row.  What the reference runs at inference with a KV cache (model.py:460-487: mask = False, the whole
// prompt against the padded cache).  N_q and N_k are independent.
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void store_one(void* base, int dtype, int64_t idx, float val) {
    if (dtype == FASTMAX_F32) reinterpret_cast<float*>(base)[idx] = val;
    else if (dtype == FASTMAX_BF16) reinterpret_cast<uint16_t*>(base)[idx] = f32_to_bf16_bits(val);
    else reinterpret_cast<_Float16*>(base)[idx] = (_Float16)val;
}
struct ApplyParams {
    const void* q;
    Strides3 qs;
    const float* total;       // [B*H] records [S2 (DP x DP) | S1 | ksum], record stride rec_stride floats
    int64_t rec_stride;
    void* o;
    float* g;
    int H, Nq, D, out_dtype;
    float a, g0;
};
// block = 256 threads = 4 waves, each wave walks rows; lane owns output column(s) lane (+ 64) and keeps that column of S2 in
// registers; the row's q values are broadcast from LDS
template <typename TIN, int DP>
__global__ __launch_bounds__(256, DP == 64 ? 2 : 1) void unmasked_p1_apply_kernel(ApplyParams prm) {
    constexpr int NC = DP / 64, RB = 128;                          // rows per block
    __shared__ float q_s[4][DP];
    __shared__ float ks_s[DP];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int bh = blockIdx.y, b = bh / prm.H, h = bh % prm.H;
    const float* rec = prm.total + (int64_t)bh * prm.rec_stride;
    float S[NC][DP], s1[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
#pragma unroll
        for (int m = 0; m < DP; ++m) S[c][m] = rec[m * DP + 64 * c + lane];
        s1[c] = rec[DP * DP + 64 * c + lane];
    }
    if (tid < DP) ks_s[tid] = rec[DP * DP + DP + tid];
    __syncthreads();
    const int row0 = blockIdx.x * RB;
    const int D = prm.D;
    for (int r = row0 + w; r < min(prm.Nq, row0 + RB); r += 4) {
        const TIN* qrow = row_ptr<TIN>(prm.q, prm.qs.sb, prm.qs.sh, prm.qs.sn, b, h, r);
        float gp = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int m = 64 * c + lane;
            const float qv = m < D ? prm.a * to_float(qrow[m]) : 0.f;
            q_s[w][m] = qv;
            gp = fmaf(qv, ks_s[m], gp);
        }
        const float gval = prm.g0 + wave_sum(gp);
        float acc[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[c] = s1[c];
        // q_s[w] was written by this wave's own lanes: a wave-level fence is enough before it is read back
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int m = 0; m < DP; ++m) {                               // fully unrolled: S stays in registers
            const float qm = q_s[w][m];                              // broadcast read
#pragma unroll
            for (int c = 0; c < NC; ++c) acc[c] = fmaf(qm, S[c][m], acc[c]);
        }
        __builtin_amdgcn_wave_barrier();
        const float inv = 1.0f / gval;
#pragma unroll
        for (int c = 0; c < NC; ++c)
            if (64 * c + lane < D) store_one(prm.o, prm.out_dtype, ((int64_t)bh * prm.Nq + r) * D + 64 * c + lane, acc[c] * inv);
        if (lane == 0 && prm.g) prm.g[(int64_t)bh * prm.Nq + r] = gval;
    }
}

// The backward of the same function, also from totals (fastmax.py:383-691 unmasked, without the O(N_q N_k) tiles):
//     w_i = 1/g_i, c_i = G_i.o_i, ghat_i = w_i G_i, e_i = -w_i c_i
//     dQ_i = a w_i S2 G_i + a e_i ksum                     S2 = sum_j k_j v_j^T, ksum = sum_j k_j          (totals over the keys)
//     dK_j = a (R2 v_j + rq)                               R2 = sum_i q_i ghat_i^T, rq = sum_i q_i e_i     (totals over the queries)
//     dV_j = R1 + a R2^T k_j                               R1 = sum_i ghat_i
// = two state passes (the forward's kernel over K, V; its reverse-scan variant over q, G with the row factors) and three
// row-wise D x D products.  out_r[c] = sb_r bias[c] + sx_r sum_m M(c, m) x_r[m]; TRANS: M(c, m) = rec[m][c], else rec[c][m].
struct Apply2Params {
    const void* x;
    Strides3 xs;
    const float* total;
    int64_t rec_stride;
    void* out;
    const float *g, *c;       // MODE 0 (dQ): row factors
    int H, N, D, out_dtype, bias_slot;    // bias_slot: 1 = the "S1" vector of the record, 2 = the "ksum" vector
    float a;
};
template <typename TIN, int DP, bool TRANS, int MODE>
__global__ __launch_bounds__(256, DP == 64 ? 2 : 1) void unmasked_p1_apply2_kernel(Apply2Params prm) {
    constexpr int NC = DP / 64, RB = 128;
    __shared__ float x_s[4][DP];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int bh = blockIdx.y, b = bh / prm.H, h = bh % prm.H;
    const float* rec = prm.total + (int64_t)bh * prm.rec_stride;
    float S[NC][DP], bias[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
#pragma unroll
        for (int m = 0; m < DP; ++m) S[c][m] = TRANS ? rec[m * DP + 64 * c + lane] : rec[(64 * c + lane) * DP + m];
        bias[c] = rec[DP * DP + (prm.bias_slot == 2 ? DP : 0) + 64 * c + lane];
    }
    const int row0 = blockIdx.x * RB, D = prm.D;
    for (int r = row0 + w; r < min(prm.N, row0 + RB); r += 4) {
        const TIN* xrow = row_ptr<TIN>(prm.x, prm.xs.sb, prm.xs.sh, prm.xs.sn, b, h, r);
        float sx = prm.a, sb = MODE == 2 ? 1.0f : prm.a;             // MODE 1 (dK): a, a;  MODE 2 (dV): a, 1
        if constexpr (MODE == 0) {                                    // dQ: a w_i, a e_i
            const float wi = 1.0f / prm.g[(int64_t)bh * prm.N + r];
            sx = prm.a * wi;
            sb = -prm.a * wi * prm.c[(int64_t)bh * prm.N + r];
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int m = 64 * c + lane;
            x_s[w][m] = m < D ? to_float(xrow[m]) : 0.f;
        }
        float acc[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[c] = 0.f;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int m = 0; m < DP; ++m) {
            const float xm = x_s[w][m];
#pragma unroll
            for (int c = 0; c < NC; ++c) acc[c] = fmaf(xm, S[c][m], acc[c]);
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int c = 0; c < NC; ++c)
            if (64 * c + lane < D) store_one(prm.out, prm.out_dtype, ((int64_t)bh * prm.N + r) * D + 64 * c + lane, fmaf(sx, acc[c], sb * bias[c]));
    }
}

static SplitPlan unmasked_plan_rows(const fastmax_problem& p, int rows);
static SplitPlan unmasked_plan(const fastmax_problem& p) { return unmasked_plan_rows(p, p.Nk); }   // segments 0 .. nseg-2 of the plan cover all keys
bool unmasked_lin_supported(const fastmax_problem& p) {
    const int epl = p.in_dtype == FASTMAX_F32 ? 4 : 8;
    // worth it once the O(N_q N_k) tiles outgrow a pass over K, V and a D x D product per query row
    return p.p == 1 && !p.causal && (p.D % epl) == 0 && p.D <= 128 && p.Nk >= 512 && p.Nq >= 64 && (int64_t)p.B * p.H <= 65535;
}
size_t unmasked_lin_workspace(const fastmax_problem& p) {
    const int dp = p.D <= 64 ? 64 : 128;
    const SplitPlan plan = unmasked_plan(p);
    return sizeof(float) * (size_t)p.B * p.H * (plan.nseg - 1) * ((size_t)dp * dp + 2 * dp);
}
int launch_fwd_unmasked_p1(const FwdArgs& a) {
    if (!unmasked_lin_supported(a.prob)) return FASTMAX_E_BAD_SHAPE;
    if (!a.workspace || a.workspace_bytes < unmasked_lin_workspace(a.prob)) return FASTMAX_E_WORKSPACE;
    const int dp = a.prob.D <= 64 ? 64 : 128;
    const SplitPlan plan = unmasked_plan(a.prob);
    FwdArgs fa = a;
    fa.prob.Nq = a.prob.Nk;                       // the state pass walks the KEY rows
    fa.stats = nullptr;
    int rc = launch_split_states(fa, plan, dp, nullptr);
    if (rc) return rc;
    const int nrec = plan.nseg - 1, rec = dp * dp + 2 * dp;
    ApplyParams prm{a.q, a.qs, reinterpret_cast<const float*>(a.workspace) + (int64_t)(nrec - 1) * rec, (int64_t)nrec * rec, a.o, a.g,
                    a.prob.H, a.prob.Nq, a.prob.D, a.prob.out_dtype, a.prob.a, a.prob.g0};
    const dim3 grid((a.prob.Nq + 127) / 128, a.prob.B * a.prob.H), block(256);
#define APPLY(T)                                                                                                   \
    if (dp == 64) hipLaunchKernelGGL((unmasked_p1_apply_kernel<T, 64>), grid, block, 0, a.stream, prm);            \
    else hipLaunchKernelGGL((unmasked_p1_apply_kernel<T, 128>), grid, block, 0, a.stream, prm)
    switch (a.prob.in_dtype) {
        case FASTMAX_F32: APPLY(float); break;
        case FASTMAX_BF16: APPLY(bf16_t); break;
        case FASTMAX_F16: APPLY(f16_t); break;
        default: return FASTMAX_E_BAD_DTYPE;
    }
#undef APPLY
    return (int)hipGetLastError();
}

static SplitPlan unmasked_plan_rows(const fastmax_problem& p, int rows) {
    const int BH = p.B * p.H, nchunks = (rows + 63) / 64;
    int nseg = (512 + BH - 1) / BH;
    if (nseg > nchunks / 4) nseg = nchunks / 4;
    if (nseg > 31) nseg = 31;
    if (nseg < 1) nseg = 1;
    const int cps = (nchunks + nseg - 1) / nseg;
    nseg = (nchunks + cps - 1) / cps;
    return SplitPlan{nseg + 1, cps};
}
static size_t align16b(size_t x) { return (x + 15) & ~(size_t)15; }
// workspace = [ c (B,H,Nq) | records over the keys | records over the queries ]
bool unmasked_lin_bwd_supported(const fastmax_problem& p) {
    // the reverse-state pass exists for D <= 64 (every dtype) and for bf16 up to 128 (as in the causal linear-time backward)
    return unmasked_lin_supported(p) && (p.D <= 64 || p.in_dtype == FASTMAX_BF16);
}
size_t unmasked_lin_bwd_workspace(const fastmax_problem& p) {
    if (!unmasked_lin_bwd_supported(p)) return 0;
    const int dp = p.D <= 64 ? 64 : 128;
    const size_t rec = sizeof(float) * ((size_t)dp * dp + 2 * dp), BH = (size_t)p.B * p.H;
    return align16b(sizeof(float) * BH * p.Nq) + align16b(BH * (unmasked_plan_rows(p, p.Nk).nseg - 1) * rec) +
           align16b(BH * (unmasked_plan_rows(p, p.Nq).nseg - 1) * rec);
}
template <typename TIN>
static void launch_apply2(const Apply2Params& prm, int dp, int mode, int BH, hipStream_t stream) {
    const dim3 grid((prm.N + 127) / 128, BH), block(256);
#define AP2(DPV, TR, MD) hipLaunchKernelGGL((unmasked_p1_apply2_kernel<TIN, DPV, TR, MD>), grid, block, 0, stream, prm)
    if (dp == 64) { if (mode == 0) AP2(64, false, 0); else if (mode == 1) AP2(64, false, 1); else AP2(64, true, 2); }
    else { if (mode == 0) AP2(128, false, 0); else if (mode == 1) AP2(128, false, 1); else AP2(128, true, 2); }
#undef AP2
}
int launch_bwd_unmasked_p1(const BwdArgs& a) {
    const fastmax_problem& p = a.prob;
    if (!unmasked_lin_bwd_supported(p)) return FASTMAX_E_BAD_SHAPE;
    if (!a.workspace || a.workspace_bytes < unmasked_lin_bwd_workspace(p)) return FASTMAX_E_WORKSPACE;
    const int dp = p.D <= 64 ? 64 : 128, BH = p.B * p.H;
    const size_t rec = (size_t)dp * dp + 2 * dp;
    const SplitPlan pk = unmasked_plan_rows(p, p.Nk), pq = unmasked_plan_rows(p, p.Nq);
    char* ws = reinterpret_cast<char*>(a.workspace);
    float* cbuf = reinterpret_cast<float*>(ws);
    float* srec = reinterpret_cast<float*>(ws + align16b(sizeof(float) * (size_t)BH * p.Nq));
    float* rrec = reinterpret_cast<float*>(reinterpret_cast<char*>(srec) + align16b(sizeof(float) * (size_t)BH * (pk.nseg - 1) * rec));
    int rc = launch_bwd_prep_c(a, cbuf);                                   // c_i = G_i . o_i
    if (rc) return rc;
    FwdArgs fa{p, a.q, a.k, a.v, a.qs, a.ks, a.vs, nullptr, nullptr, srec, sizeof(float) * (size_t)BH * (pk.nseg - 1) * rec, a.stream};
    fa.prob.Nq = p.Nk;                                                     // totals over the key rows
    rc = launch_split_states(fa, pk, dp, nullptr);
    if (rc) return rc;
    // totals over the query rows: R2 = sum q ghat^T, R1 = sum ghat, rq = sum q e (every segment, suffix-summed: record 0)
    rc = launch_split_rstates(a.q, a.qs, a.grad_o, a.gos, a.g, cbuf, rrec, p, pq, dp, a.stream, nullptr, 0);
    if (rc) return rc;
    const float* stot = srec + (size_t)(pk.nseg - 2) * rec;
    Apply2Params pdq{a.grad_o, a.gos, stot, (int64_t)(pk.nseg - 1) * (int64_t)rec, a.dq, a.g, cbuf, p.H, p.Nq, p.D, p.in_dtype, 2, p.a};
    Apply2Params pdk{a.v, a.vs, rrec, (int64_t)(pq.nseg - 1) * (int64_t)rec, a.dk, nullptr, nullptr, p.H, p.Nk, p.D, p.in_dtype, 2, p.a};
    Apply2Params pdv{a.k, a.ks, rrec, (int64_t)(pq.nseg - 1) * (int64_t)rec, a.dv, nullptr, nullptr, p.H, p.Nk, p.D, p.in_dtype, 1, p.a};
    switch (p.in_dtype) {
        case FASTMAX_F32: launch_apply2<float>(pdq, dp, 0, BH, a.stream); launch_apply2<float>(pdk, dp, 1, BH, a.stream); launch_apply2<float>(pdv, dp, 2, BH, a.stream); break;
        case FASTMAX_BF16: launch_apply2<bf16_t>(pdq, dp, 0, BH, a.stream); launch_apply2<bf16_t>(pdk, dp, 1, BH, a.stream); launch_apply2<bf16_t>(pdv, dp, 2, BH, a.stream); break;
        case FASTMAX_F16: launch_apply2<f16_t>(pdq, dp, 0, BH, a.stream); launch_apply2<f16_t>(pdk, dp, 1, BH, a.stream); launch_apply2<f16_t>(pdv, dp, 2, BH, a.stream); break;
        default: return FASTMAX_E_BAD_DTYPE;
    }
    return (int)hipGetLastError();
}

}  // namespace fastmax
