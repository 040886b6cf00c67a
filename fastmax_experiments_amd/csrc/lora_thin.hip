// The low-rank branch of the QLoRA linear at training sizes (reference: lit_gpt/lora.py:170-177 LoRALinear.forward,
// :398-433 LoRAQKVLinear.forward, and the autograd mirror of those lines).  With the frozen base product on the library
// GEMM (M >= 2048 rows), what is left of the layer are rank-r products that each stream one (M x K) or (M x N) bf16
// matrix once -- HBM-bound passes, one kernel each:
//     down : E[M][RP]   = X[M][K] . Bt[RP][K]^T            ea = x A^T          d_ea = dy eb
//            (+ E^T, zero padded, for the next kernel)
//     tn   : C[RP][N]   = E^T[RP][M] . X[M][N]   (fp32)    d_eb^T = ea^T dy    dA = d_ea^T x
//     up   : Y[M][N]   += E[M][R] . Bn[N][R]^T (+ bias)    y += ea eb^T        dx += d_ea A
// All operands bf16, fp32 accumulation.  RP = rank padded to 16 or 32 (zero columns).
//   down: one workgroup per 16 rows, the 4 waves split K in interleaved 64-column steps (two 16x16x32 MFMA k-steps, each
//         load instruction takes 64 contiguous bytes of 16 rows), cross-wave sum through LDS.
//   tn  : the contraction runs down the rows of X, so 64-row x 64-column tiles are staged in an LDS image and read with
//         ds_read_b64_tr_b16 (the same transposed fragment the fastmax state kernels use); workgroups own a 64-column slab
//         and a range of rows; X tiles and the matching E^T columns are fetched two stages ahead and staged one stage ahead;
//         per-range partials are summed (and cast / transposed) by a second small kernel in a fixed order.
//   up  : read-modify-write of Y with 16-byte accesses, lane = 8 (or 4) consecutive columns, wave = one row per step;
//         the row's R coefficients are wave-uniform (one dword per lane + v_readlane), Bn lives in registers.  As many waves
//         as the device holds at once, each walking its column slab with two row blocks in flight.
//         16 R FMAs per 16 bytes: VALU, under the HBM time.
#include <stdlib.h>

#include "fastmax_mfma_common.h"

namespace fastmax {

typedef unsigned int tu32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int tu32x2 __attribute__((ext_vector_type(2)));

// plain loads: every matrix streamed here is read again by the next kernel of the layer (measured: with non-temporal
// loads of y the RoPE pass that follows ran 15 us slower)
#ifdef THIN_NT
#define THIN_STREAM_LOAD(p) __builtin_nontemporal_load(p)
#else
#define THIN_STREAM_LOAD(p) (*(p))
#endif

// ---------------------------------------------------------------------------------------------------------------------
// LoRA dropout (lit_gpt/lora.py:175, 422: `self.lora_dropout(x)` on the branch input, finetune/lora.py:42 lora_dropout = 0.05)
// inside the rank-r kernels: the keep mask of x (M x K) is never stored.  It is a counter-based function of
// (seed, row, column), so the forward pass (down: x A^T) and the backward pass (tn: dA = d_ea^T dropout(x); up: dx += mask o
// (d_ea A)) regenerate the same bits.  Element (m, k): 16 bits of lowbias32(seed ^ (m * ldw + k / 2)) -- the low half for an
// even k, the high half for an odd one -- kept iff >= thresh = round(65536 p); kept values are scaled by 65536 / (65536 -
// thresh) on the product side (the accumulators, not the operands).  The seed is read from device memory, so that a
// captured HIP graph draws a fresh mask on every replay.
// ---------------------------------------------------------------------------------------------------------------------
struct DropParams {
    const unsigned int* seed;     // device pointer (2 words), null = no dropout
    unsigned int thresh;          // of 65536
    unsigned int ldw;             // words (pairs of columns) per mask row
    float scale;                  // 1 / keep probability
};
__device__ __forceinline__ unsigned int drop_hash(unsigned int seed, unsigned int row, unsigned int word, unsigned int ldw) {
    unsigned int x = seed ^ (row * ldw + word);
    x ^= x >> 16; x *= 0x21f0aaadu; x ^= x >> 15; x *= 0x735a2d97u; x ^= x >> 15;
    return x;
}
// and-mask for the two bf16 values of word `word` of mask row `row`
__device__ __forceinline__ unsigned int drop_word_mask(unsigned int seed, unsigned int row, unsigned int word, const DropParams& d) {
    const unsigned int h = drop_hash(seed, row, word, d.ldw);
    return ((h & 0xffffu) >= d.thresh ? 0x0000ffffu : 0u) | ((h >> 16) >= d.thresh ? 0xffff0000u : 0u);
}
// zero the dropped elements of a 16-byte piece that starts at even column col0 of row
__device__ __forceinline__ tu32x4 drop_piece(tu32x4 v, unsigned int seed, int row, int col0, const DropParams& d) {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] &= drop_word_mask(seed, (unsigned int)row, (unsigned int)(col0 >> 1) + i, d);
    return v;
}

// ---------------------------------------------------------------------------------------------------------------------
// down
// ---------------------------------------------------------------------------------------------------------------------
struct DownParams {
    const __bf16* x; int64_t ldx;
    const __bf16* bt; int64_t ldbt;
    __bf16* e; int64_t lde;
    __bf16* et; int64_t ldet;      // may be null
    int M, K;
    DropParams drop;
};

template <int RPB, bool DROP = false>   // RP = 16 RPB; DROP: x passes through the dropout mask
__global__ __launch_bounds__(256) void lora_down_kernel(const DownParams p) {
    __shared__ float red[4][RPB][4][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
    const int m0 = blockIdx.x * 16;
    const int m = m0 + r;
    const bool live = m < p.M;
    const __bf16* xr = p.x + (int64_t)(live ? m : 0) * p.ldx + 8 * q;
    const __bf16* br = p.bt + (int64_t)r * p.ldbt + 8 * q;
    f32x4 acc[RPB];
#pragma unroll
    for (int cb = 0; cb < RPB; ++cb) acc[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nstep = p.K >> 6;
    unsigned int seed = 0;
    if constexpr (DROP) seed = p.drop.seed[0];
    struct Step { tu32x4 x0, x1; bf16x8 b0[RPB], b1[RPB]; int kb; };
    auto fetch = [&](int d, Step& t) {
        const int kb = d << 6;
        t.kb = kb;
        t.x0 = tu32x4{0, 0, 0, 0};
        t.x1 = tu32x4{0, 0, 0, 0};
        if (live) {
            t.x0 = THIN_STREAM_LOAD(reinterpret_cast<const tu32x4*>(xr + kb));
            t.x1 = THIN_STREAM_LOAD(reinterpret_cast<const tu32x4*>(xr + kb + 32));
        }
#pragma unroll
        for (int cb = 0; cb < RPB; ++cb) {
            t.b0[cb] = *reinterpret_cast<const bf16x8*>(br + (int64_t)cb * 16 * p.ldbt + kb);
            t.b1[cb] = *reinterpret_cast<const bf16x8*>(br + (int64_t)cb * 16 * p.ldbt + kb + 32);
        }
    };
    auto consume = [&](const Step& t) {
        tu32x4 x0 = t.x0, x1 = t.x1;
        if constexpr (DROP) {                       // the mask is applied when the piece is consumed: the hashing overlaps the loads
            x0 = drop_piece(x0, seed, m, t.kb + 8 * q, p.drop);
            x1 = drop_piece(x1, seed, m, t.kb + 32 + 8 * q, p.drop);
        }
#pragma unroll
        for (int cb = 0; cb < RPB; ++cb) {
            acc[cb] = mfma(__builtin_bit_cast(bf16x8, x0), t.b0[cb], acc[cb]);
            acc[cb] = mfma(__builtin_bit_cast(bf16x8, x1), t.b1[cb], acc[cb]);
        }
    };
    // this wave's steps are wave, wave + 4, ...; the sweep starts at a column that depends on the workgroup so that the
    // resident workgroups do not all walk the same columns (the same memory channels) at the same time
    const int nj = (nstep - wave + 3) >> 2;
    const int rot = nj > 0 ? (int)(blockIdx.x % (unsigned)nj) : 0;
    auto col = [&](int j) { int jj = j + rot; if (jj >= nj) jj -= nj; return wave + 4 * jj; };
    int j = 0;
    for (; j + 3 < nj; j += 4) {               // four steps of this wave in flight
        Step t0, t1, t2, t3;
        fetch(col(j), t0);
        fetch(col(j + 1), t1);
        fetch(col(j + 2), t2);
        fetch(col(j + 3), t3);
        consume(t0);
        consume(t1);
        consume(t2);
        consume(t3);
    }
    for (; j < nj; ++j) {
        Step t;
        fetch(col(j), t);
        consume(t);
    }
#pragma unroll
    for (int cb = 0; cb < RPB; ++cb)
#pragma unroll
        for (int i = 0; i < 4; ++i) red[wave][cb][i][lane] = acc[cb][i];
    __syncthreads();
    {
        const int i = tid >> 6, l = tid & 63;           // accumulator register i of lane l: row 4 (l>>4) + i, column l & 15
        const int row = m0 + 4 * (l >> 4) + i, col = l & 15;
#pragma unroll
        for (int cb = 0; cb < RPB; ++cb) {
            float s = (red[0][cb][i][l] + red[1][cb][i][l]) + (red[2][cb][i][l] + red[3][cb][i][l]);
            if constexpr (DROP) s *= p.drop.scale;
            const __bf16 v = (__bf16)s;
            if (row < p.M) p.e[(int64_t)row * p.lde + cb * 16 + col] = v;
            if (p.et) p.et[(int64_t)(cb * 16 + col) * p.ldet + row] = row < p.M ? v : (__bf16)0.f;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// tn
// ---------------------------------------------------------------------------------------------------------------------
struct TnParams {
    const __bf16* et; int64_t ldet; int etcols;    // columns >= M up to etcols are zero; etcols % 4 == 0
    const __bf16* x; int64_t ldx;
    float* part;                                   // (S, RP, ncols)
    int M, ncols, rps;                             // rows per split, a whole number of stages
    DropParams drop;                               // DROP: X is dropout(x), the mask regenerated per piece (the scale is applied by the caller's reduce pass)
};

template <int RPB, int KS, bool DROP = false>   // stages of 32 KS rows
__global__ __launch_bounds__(256) void lora_tn_kernel(const TnParams p) {
    constexpr int SR = 32 * KS, SB = SR * 128;                  // X stage: SR rows x 64 columns
    constexpr int RSA = 2 * SR + 16, AB = 16 * RPB * RSA;       // E^T stage: 16 RPB rows x SR columns, padded rows
    constexpr int NP8 = (RPB * KS + 1) / 2;                     // 8-byte pieces of the E^T stage per thread
    __shared__ __attribute__((aligned(16))) char smem[2 * SB + 2 * AB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
    const int n0 = blockIdx.x * 64, s = blockIdx.y;
    const int mbeg = s * p.rps, mend = min(p.M, mbeg + p.rps);
    const int nst = (mend - mbeg + SR - 1) / SR;
    f32x4 acc[RPB];
#pragma unroll
    for (int cb = 0; cb < RPB; ++cb) acc[cb] = f32x4{0.f, 0.f, 0.f, 0.f};

    // both operands go global -> registers (two stages ahead) -> LDS (one stage ahead) -> fragments
    struct Regs { tu32x4 x[KS]; tu32x2 a[NP8]; };
    auto fetch = [&](int st, Regs& t) {
        const int mb = mbeg + st * SR;
#pragma unroll
        for (int i = 0; i < KS; ++i) {
            const int piece = tid + 256 * i, row = piece >> 3, chunk = piece & 7;
            t.x[i] = tu32x4{0, 0, 0, 0};
            if (mb + row < mend) t.x[i] = THIN_STREAM_LOAD(reinterpret_cast<const tu32x4*>(p.x + (int64_t)(mb + row) * p.ldx + n0 + 8 * chunk));
        }
#pragma unroll
        for (int i = 0; i < NP8; ++i) {
            const int piece = tid + 256 * i, c = piece / (8 * KS), mm = mb + (piece % (8 * KS)) * 4;
            t.a[i] = tu32x2{0, 0};
            if (c < 16 * RPB && mm < mend && mm + 4 <= p.etcols) t.a[i] = *reinterpret_cast<const tu32x2*>(p.et + (int64_t)c * p.ldet + mm);
        }
    };
    unsigned int seed = 0;
    if constexpr (DROP) seed = p.drop.seed[0];
    auto stage = [&](const Regs& t, int buf, int st) {
#pragma unroll
        for (int i = 0; i < KS; ++i) {
            const int piece = tid + 256 * i, row = piece >> 3, chunk = piece & 7;
            tu32x4 xv = t.x[i];
            if constexpr (DROP) xv = drop_piece(xv, seed, mbeg + st * SR + row, n0 + 8 * chunk, p.drop);
            *reinterpret_cast<tu32x4*>(smem + buf * SB + img_off<64>(row, chunk)) = xv;
        }
#pragma unroll
        for (int i = 0; i < NP8; ++i) {
            const int piece = tid + 256 * i, c = piece / (8 * KS), mo = (piece % (8 * KS)) * 4;
            if (c < 16 * RPB) *reinterpret_cast<tu32x2*>(smem + 2 * SB + buf * AB + c * RSA + mo * 2) = t.a[i];
        }
    };
    auto step = [&](int st, const Regs& next, Regs& free) {
        const int buf = st & 1;
        if (st + 2 < nst) fetch(st + 2, free);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bf16x8 b = ld_tr8<64>(smem, buf * SB, 32 * ks, 16 * wave, lane);
#pragma unroll
            for (int cb = 0; cb < RPB; ++cb) {
                // the k order of ld_tr8: rows 4q+{0..3}, then 16+4q+{0..3}
                const char* ap = smem + 2 * SB + buf * AB + (cb * 16 + r) * RSA + (32 * ks + 4 * q) * 2;
                union { bf16x8 v; tu32x2 h[2]; } u;
                u.h[0] = *reinterpret_cast<const tu32x2*>(ap);
                u.h[1] = *reinterpret_cast<const tu32x2*>(ap + 32);
                acc[cb] = mfma(u.v, b, acc[cb]);
            }
        }
        if (st + 1 < nst) stage(next, buf ^ 1, st + 1);
        __syncthreads();
    };
    Regs ra, rb;
    if (nst > 0) {
        fetch(0, ra);
        if (nst > 1) fetch(1, rb);
        stage(ra, 0, 0);
    }
    __syncthreads();
    for (int st = 0; st < nst; st += 2) {
        step(st, rb, ra);
        if (st + 1 < nst) step(st + 1, ra, rb);
    }
    // C: column n = 16 wave + (lane & 15), row c = 16 cb + 4 q + i
    float* out = p.part + ((int64_t)s * RPB * 16) * p.ncols + n0 + 16 * wave + r;
#pragma unroll
    for (int cb = 0; cb < RPB; ++cb)
#pragma unroll
        for (int i = 0; i < 4; ++i) out[(int64_t)(cb * 16 + 4 * q + i) * p.ncols] = acc[cb][i];
}

// out[c][n] (or out[n][c] when transposed) = sum_s part[s][c][n] for c < R, summed in split order; float32 or bf16
__global__ __launch_bounds__(256) void lora_tn_reduce_kernel(const float* part, void* out, int S, int RP, int R, int ncols, int out_bf16,
                                                             int transpose, float scale) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)R * ncols) return;
    const int c = (int)(i / ncols), n = (int)(i % ncols);
    const int64_t plane = (int64_t)RP * ncols;
    const float* src = part + (int64_t)c * ncols + n;
    float s = 0.f;
    int k = 0;
    for (; k + 8 <= S; k += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[(int64_t)(k + u) * plane];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; k < S; ++k) s += src[(int64_t)k * plane];
    s *= scale;                                                                          // 1 / keep probability under dropout, else 1
    const int64_t o = transpose ? (int64_t)n * R + c : i;
    if (out_bf16) reinterpret_cast<__bf16*>(out)[o] = (__bf16)s;
    else reinterpret_cast<float*>(out)[o] = s;
}

static int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}
static int tn_ks() { static const int ks = env_int("FASTMAX_LORA_TN_KS", 2) == 4 ? 4 : 2; return ks; }

static void tn_plan(int M, int ncols, int& S, int& rps) {
    static const int target = env_int("FASTMAX_LORA_TN_TARGET", 1024);
    const int slabs = ncols / 64, sr = 32 * tn_ks();
    int want = (target + slabs - 1) / slabs;
    const int maxs = (M + sr - 1) / sr;
    if (want > maxs) want = maxs;
    if (want < 1) want = 1;
    rps = (((M + want - 1) / want) + sr - 1) / sr * sr;
    S = (M + rps - 1) / rps;
}

// ---------------------------------------------------------------------------------------------------------------------
// up
// ---------------------------------------------------------------------------------------------------------------------
struct UpParams {
    __bf16* y; int64_t ldy;
    const __bf16* e; int64_t lde;
    const __bf16* bn; int64_t ldb;  // (N, R) rows, or (R, N) when bn_t
    const float* bias;             // may be null
    int bn_t;
    int M, N, nslab, groups;       // groups: waves per column slab
    DropParams drop;               // DROP: y[m][n] += keep(m, n) / (1 - p) * (e bn^T)[m][n]   (dx of the dropped-out LoRA input)
};

template <int NPL> struct UpVec;
template <> struct UpVec<8> { typedef tu32x4 type; };
template <> struct UpVec<4> { typedef tu32x2 type; };

__device__ __forceinline__ float bf_lo(unsigned int w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(unsigned int w) { return __uint_as_float(w & 0xffff0000u); }

// Each wave keeps one slab of 64 NPL columns (its Bn rows stay in registers) and walks row blocks of RU rows with a stride
// of `groups` blocks, two blocks in flight: the loads of block i+1 are issued before block i is finished.  The block's
// RU x R coefficients are fetched with one dword per lane and broadcast with v_readlane.
template <int R, int NPL, int RU, bool DROP = false>
__global__ __launch_bounds__(256) void lora_up_kernel(const UpParams p) {
    typedef typename UpVec<NPL>::type vec_t;
    static_assert(RU * R / 2 <= 64, "one dword of E per lane");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int w = blockIdx.x * 4 + wave;
    if (w >= p.nslab * p.groups) return;
    const int slab = w % p.nslab, g = w / p.nslab;
    const int n = slab * 64 * NPL + lane * NPL;
    const bool act = n < p.N;
    const int nn = act ? n : 0;                                              // idle lanes read column 0 and store nothing
    const int nblk = (p.M + RU - 1) / RU;
    float b[NPL][R], bias[NPL];
#pragma unroll
    for (int j = 0; j < NPL; ++j) bias[j] = p.bias ? p.bias[nn + j] : 0.f;
    if (p.bn_t) {                                                            // (R, N): NPL consecutive columns of row c
#pragma unroll
        for (int c = 0; c < R; ++c) {
            const vec_t wv = *reinterpret_cast<const vec_t*>(p.bn + (int64_t)c * p.ldb + nn);
#pragma unroll
            for (int j = 0; j < NPL; j += 2) {
                b[j][c] = bf_lo(wv[j >> 1]);
                b[j + 1][c] = bf_hi(wv[j >> 1]);
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
#pragma unroll
            for (int c = 0; c < R; c += 8) {
                const tu32x4 wv = *reinterpret_cast<const tu32x4*>(p.bn + (int64_t)(nn + j) * p.ldb + c);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    b[j][c + 2 * t] = bf_lo(wv[t]);
                    b[j][c + 2 * t + 1] = bf_hi(wv[t]);
                }
            }
        }
    }
    struct Blk { vec_t y[RU]; unsigned int e; };
    unsigned int seed = 0;
    if constexpr (DROP) seed = p.drop.seed[0];
    const int eu = lane / (R / 2), ec = lane % (R / 2);                      // this lane's dword of the block's coefficients
    auto fetch = [&](int bi, Blk& t) {
        const int m = bi * RU;
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            t.y[u] = vec_t{};
            if (act && m + u < p.M) t.y[u] = THIN_STREAM_LOAD(reinterpret_cast<const vec_t*>(p.y + (int64_t)(m + u) * p.ldy + n));
        }
        t.e = 0;
        if (eu < RU && m + eu < p.M) t.e = *reinterpret_cast<const unsigned int*>(p.e + (int64_t)(m + eu) * p.lde + 2 * ec);
    };
    auto finish = [&](int bi, const Blk& t) {
        const int m = bi * RU;
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            float o[NPL], d[NPL];
#pragma unroll
            for (int j = 0; j < NPL; j += 2) {
                const unsigned int wv = t.y[u][j >> 1];
                o[j] = bf_lo(wv) + bias[j];
                o[j + 1] = bf_hi(wv) + bias[j + 1];
                d[j] = d[j + 1] = 0.f;
            }
#pragma unroll
            for (int c = 0; c < R; c += 2) {
                const unsigned int wv = __builtin_amdgcn_readlane(t.e, u * (R / 2) + (c >> 1));
                const float e0 = bf_lo(wv), e1 = bf_hi(wv);
#pragma unroll
                for (int j = 0; j < NPL; ++j) {
                    if constexpr (DROP) d[j] = fmaf(e1, b[j][c + 1], fmaf(e0, b[j][c], d[j]));
                    else o[j] = fmaf(e1, b[j][c + 1], fmaf(e0, b[j][c], o[j]));
                }
            }
            if constexpr (DROP) {
#pragma unroll
                for (int j = 0; j < NPL; j += 2) {
                    const unsigned int km = drop_word_mask(seed, (unsigned int)(m + u), (unsigned int)((n + j) >> 1), p.drop);
                    o[j] += (km & 0xffffu) ? d[j] * p.drop.scale : 0.f;
                    o[j + 1] += (km >> 16) ? d[j + 1] * p.drop.scale : 0.f;
                }
            }
            vec_t ov;
#pragma unroll
            for (int j = 0; j < NPL; j += 2)
                ov[j >> 1] = (unsigned int)f32_to_bf16_bits(o[j]) | ((unsigned int)f32_to_bf16_bits(o[j + 1]) << 16);
            if (act && m + u < p.M) *reinterpret_cast<vec_t*>(p.y + (int64_t)(m + u) * p.ldy + n) = ov;
        }
    };
    const int G = p.groups;
    Blk ba, bb;
    int bi = g;
    if (bi < nblk) fetch(bi, ba);
    for (; bi < nblk; bi += 2 * G) {
        if (bi + G < nblk) fetch(bi + G, bb);
        finish(bi, ba);
        if (bi + 2 * G < nblk) fetch(bi + 2 * G, ba);
        if (bi + G < nblk) finish(bi + G, bb);
    }
}

template <int R, int NPL, int RU> static int up_launch(UpParams p, hipStream_t stream) {
    static int resident = 0;                          // waves of this kernel the device holds at once
    if (!resident) {
        int per_cu = 0, dev = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, lora_up_kernel<R, NPL, RU, false>, 256, 0) != hipSuccess || per_cu < 1) per_cu = 1;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
        resident = per_cu * cus * 4 * env_int("FASTMAX_LORA_UP_OVERSUB", 1);
    }
    p.nslab = (p.N + 64 * NPL - 1) / (64 * NPL);
    const int nblk = (p.M + RU - 1) / RU;
    int groups = resident / p.nslab;
    if (groups < 1) groups = 1;
    if (groups > nblk) groups = nblk;
    p.groups = groups;
    const int64_t waves = (int64_t)p.nslab * groups;
    if (p.drop.seed) hipLaunchKernelGGL((lora_up_kernel<R, NPL, RU, true>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((lora_up_kernel<R, NPL, RU, false>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, stream, p);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// the (RP, N) operand of the LoRA branch from lora_B: E^T[part r + j][n] = scaling B[row(part, n)][j], zero where the
// output column n has no row in that part (LoRAQKVLinear's zero_pad / lora_ind scatter, lit_gpt/lora.py:263-342) and in
// the padding rows; and the gather back for the gradient of lora_B
// ---------------------------------------------------------------------------------------------------------------------
template <typename TB>
__global__ __launch_bounds__(256) void lora_scatter_kernel(const TB* b, int r, const int* rowmap, int n_parts, float scaling, __bf16* et,
                                                           int64_t ldet, int N, int RP) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)RP * N) return;
    const int c = (int)(i / N), n = (int)(i % N);
    const int part = c / r, j = c % r;
    float v = 0.f;
    if (part < n_parts) {
        const int row = rowmap[(int64_t)part * N + n];
        if (row >= 0) v = scaling * to_float(b[(int64_t)row * r + j]);
    }
    et[(int64_t)c * ldet + n] = (__bf16)v;
}

template <typename TB, typename TG>
__global__ __launch_bounds__(256) void lora_scatter_bwd_kernel(const TG* d_et, int64_t ldd, const int* ind, const int* part, float scaling, TB* db,
                                                               int n_rows, int r) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)n_rows * r) return;
    const int row = (int)(i / r), j = (int)(i % r);
    db[i] = from_float<TB>(scaling * to_float(d_et[(int64_t)(part[row] * r + j) * ldd + ind[row]]));
}

static bool aligned16(const void* ptr, int64_t ld_elems) { return !(reinterpret_cast<uintptr_t>(ptr) & 15) && (ld_elems * 2) % 16 == 0; }

}  // namespace fastmax

using namespace fastmax;

extern "C" {

static DropParams make_drop(const void* seed, float p_drop, int cols) {
    DropParams d{nullptr, 0u, (unsigned int)(cols / 2), 1.0f};
    if (seed && p_drop > 0.f) {
        unsigned int t = (unsigned int)(p_drop * 65536.0f + 0.5f);
        if (t > 65535u) t = 65535u;
        d.seed = reinterpret_cast<const unsigned int*>(seed);
        d.thresh = t;
        d.scale = 65536.0f / (float)(65536u - t);
    }
    return d;
}

static int lora_down_impl(const void* x, int64_t ldx, const void* bt, int64_t ldbt, void* e, int64_t lde, void* et, int64_t ldet, int M,
                          int K, int RP, const void* seed, float p_drop, void* stream) {
    if (!x || !bt || !e) return FASTMAX_E_NULL;
    if (M <= 0 || K <= 0 || K % 64 || (RP != 16 && RP != 32) || lde < RP || ldx < K || ldbt < K) return FASTMAX_E_BAD_SHAPE;
    if (!aligned16(x, ldx) || !aligned16(bt, ldbt)) return FASTMAX_E_BAD_SHAPE;
    int rows = M;
    if (et) {
        if (ldet < (M + 15) / 16 * 16) return FASTMAX_E_BAD_SHAPE;
        rows = (int)(ldet / 16 * 16);                   // the transposed copy is zero filled up to its leading dimension
    }
    DownParams p{reinterpret_cast<const __bf16*>(x), ldx, reinterpret_cast<const __bf16*>(bt), ldbt, reinterpret_cast<__bf16*>(e), lde,
                 reinterpret_cast<__bf16*>(et), ldet, M, K, make_drop(seed, p_drop, K)};
    const dim3 grid((rows + 15) / 16);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (p.drop.seed) {
        if (RP == 16) hipLaunchKernelGGL((lora_down_kernel<1, true>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((lora_down_kernel<2, true>), grid, dim3(256), 0, s, p);
    } else {
        if (RP == 16) hipLaunchKernelGGL((lora_down_kernel<1, false>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((lora_down_kernel<2, false>), grid, dim3(256), 0, s, p);
    }
    return (int)hipGetLastError();
}

int fastmax_hip_lora_down(const void* x, int64_t ldx, const void* bt, int64_t ldbt, void* e, int64_t lde, void* et,
                          int64_t ldet, int M, int K, int RP, void* stream) {
    return lora_down_impl(x, ldx, bt, ldbt, e, lde, et, ldet, M, K, RP, nullptr, 0.f, stream);
}
int fastmax_hip_lora_down_dropout(const void* x, int64_t ldx, const void* bt, int64_t ldbt, void* e, int64_t lde, void* et,
                                  int64_t ldet, int M, int K, int RP, const void* seed, float p_drop, void* stream) {
    return lora_down_impl(x, ldx, bt, ldbt, e, lde, et, ldet, M, K, RP, seed, p_drop, stream);
}

int64_t fastmax_hip_lora_tn_workspace(int M, int ncols, int RP) {
    if (M <= 0 || ncols <= 0 || ncols % 64 || (RP != 16 && RP != 32)) return -1;
    int S, rps;
    tn_plan(M, ncols, S, rps);
    return (int64_t)S * RP * ncols * 4;
}

static int lora_tn_impl(const void* et, int64_t ldet, const void* x, int64_t ldx, void* out, int out_dtype, int transpose, int R,
                        void* workspace, int M, int ncols, int RP, const void* seed, float p_drop, void* stream) {
    if (!et || !x || !out || !workspace) return FASTMAX_E_NULL;
    if (M <= 0 || ncols <= 0 || ncols % 64 || (RP != 16 && RP != 32) || ldx < ncols || R <= 0 || R > RP) return FASTMAX_E_BAD_SHAPE;
    if (out_dtype != FASTMAX_F32 && out_dtype != FASTMAX_BF16) return FASTMAX_E_BAD_DTYPE;
    const int etcols = (M + 15) / 16 * 16;
    if (ldet < etcols || (ldet * 2) % 8 || (reinterpret_cast<uintptr_t>(et) & 7) || !aligned16(x, ldx)) return FASTMAX_E_BAD_SHAPE;
    int S, rps;
    tn_plan(M, ncols, S, rps);
    TnParams p{reinterpret_cast<const __bf16*>(et), ldet, etcols, reinterpret_cast<const __bf16*>(x), ldx, reinterpret_cast<float*>(workspace),
               M, ncols, rps, make_drop(seed, p_drop, ncols)};
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid(ncols / 64, S);
#define TN_LAUNCH(RPB, KS)                                                                                      \
    do {                                                                                                        \
        if (p.drop.seed) hipLaunchKernelGGL((lora_tn_kernel<RPB, KS, true>), grid, dim3(256), 0, s, p);         \
        else hipLaunchKernelGGL((lora_tn_kernel<RPB, KS, false>), grid, dim3(256), 0, s, p);                    \
    } while (0)
    if (tn_ks() == 4) {
        if (RP == 16) TN_LAUNCH(1, 4);
        else TN_LAUNCH(2, 4);
    } else {
        if (RP == 16) TN_LAUNCH(1, 2);
        else TN_LAUNCH(2, 2);
    }
#undef TN_LAUNCH
    const int64_t n = (int64_t)R * ncols;
    hipLaunchKernelGGL(lora_tn_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, reinterpret_cast<const float*>(workspace), out, S,
                       RP, R, ncols, out_dtype == FASTMAX_BF16, transpose, p.drop.scale);
    return (int)hipGetLastError();
}
int fastmax_hip_lora_tn(const void* et, int64_t ldet, const void* x, int64_t ldx, void* out, int out_dtype, int transpose, int R,
                        void* workspace, int M, int ncols, int RP, void* stream) {
    return lora_tn_impl(et, ldet, x, ldx, out, out_dtype, transpose, R, workspace, M, ncols, RP, nullptr, 0.f, stream);
}
int fastmax_hip_lora_tn_dropout(const void* et, int64_t ldet, const void* x, int64_t ldx, void* out, int out_dtype, int transpose, int R,
                                void* workspace, int M, int ncols, int RP, const void* seed, float p_drop, void* stream) {
    return lora_tn_impl(et, ldet, x, ldx, out, out_dtype, transpose, R, workspace, M, ncols, RP, seed, p_drop, stream);
}

static int lora_up_impl(void* y, int64_t ldy, const void* e, int64_t lde, const void* bn, int64_t ldb, int bn_transposed,
                        const float* bias, int M, int N, int R, const void* seed, float p_drop, void* stream) {
    if (!y || !e || !bn) return FASTMAX_E_NULL;
    if (M <= 0 || N <= 0 || N % 8 || ldy < N || lde < R || ldb < (bn_transposed ? N : R)) return FASTMAX_E_BAD_SHAPE;
    if (!aligned16(y, ldy) || !aligned16(e, lde) || !aligned16(bn, ldb)) return FASTMAX_E_BAD_SHAPE;
    UpParams p{reinterpret_cast<__bf16*>(y), ldy, reinterpret_cast<const __bf16*>(e), lde, reinterpret_cast<const __bf16*>(bn), ldb, bias, bn_transposed ? 1 : 0, M, N, 0, 0,
               make_drop(seed, p_drop, N)};
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    static const int variant = env_int("FASTMAX_LORA_UP_VARIANT", 0);
    switch (R) {
        case 8: return variant & 1 ? up_launch<8, 8, 8>(p, s) : up_launch<8, 8, 4>(p, s);
        case 16:
            switch (variant) {
                case 1: return up_launch<16, 4, 8>(p, s);
                case 2: return up_launch<16, 4, 4>(p, s);
                case 3: return up_launch<16, 8, 8>(p, s);
                default: return up_launch<16, 8, 4>(p, s);
            }
        case 24: return up_launch<24, 4, 4>(p, s);
        case 32: return up_launch<32, 4, 4>(p, s);
    }
    return FASTMAX_E_BAD_SHAPE;
}
int fastmax_hip_lora_up(void* y, int64_t ldy, const void* e, int64_t lde, const void* bn, int64_t ldb, int bn_transposed,
                        const float* bias, int M, int N, int R, void* stream) {
    return lora_up_impl(y, ldy, e, lde, bn, ldb, bn_transposed, bias, M, N, R, nullptr, 0.f, stream);
}
int fastmax_hip_lora_up_dropout(void* y, int64_t ldy, const void* e, int64_t lde, const void* bn, int64_t ldb, int bn_transposed,
                                const float* bias, int M, int N, int R, const void* seed, float p_drop, void* stream) {
    return lora_up_impl(y, ldy, e, lde, bn, ldb, bn_transposed, bias, M, N, R, seed, p_drop, stream);
}

// the keep mask itself (1 = kept), M x K bytes: what the three kernels above regenerate on the fly (tests, inspection)
__global__ __launch_bounds__(256) void lora_dropout_mask_kernel(unsigned char* mask, int M, int K, fastmax::DropParams d) {
    const int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int ldw = K / 2;
    if (w >= (int64_t)M * ldw) return;
    const int row = (int)(w / ldw), word = (int)(w % ldw);
    const unsigned int km = fastmax::drop_word_mask(d.seed[0], (unsigned int)row, (unsigned int)word, d);
    mask[(int64_t)row * K + 2 * word] = (km & 0xffffu) ? 1 : 0;
    mask[(int64_t)row * K + 2 * word + 1] = (km >> 16) ? 1 : 0;
}
int fastmax_hip_lora_dropout_mask(void* mask, int M, int K, const void* seed, float p_drop, void* stream) {
    if (!mask || !seed) return FASTMAX_E_NULL;
    if (M <= 0 || K <= 0 || K % 2 || !(p_drop > 0.f)) return FASTMAX_E_BAD_SHAPE;
    const int64_t n = (int64_t)M * (K / 2);
    hipLaunchKernelGGL(lora_dropout_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<unsigned char*>(mask), M, K, make_drop(seed, p_drop, K));
    return (int)hipGetLastError();
}

int fastmax_hip_lora_scatter(const void* b, int b_dtype, int r, const int32_t* rowmap, int n_parts, float scaling, void* et,
                             int64_t ldet, int N, int RP, void* stream) {
    if (!b || !rowmap || !et) return FASTMAX_E_NULL;
    if (r <= 0 || n_parts <= 0 || n_parts * r > RP || N <= 0 || ldet < N) return FASTMAX_E_BAD_SHAPE;
    const int64_t n = (int64_t)RP * N;
    const dim3 grid((unsigned)((n + 255) / 256));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (b_dtype == FASTMAX_F32)
        hipLaunchKernelGGL((lora_scatter_kernel<float>), grid, dim3(256), 0, s, reinterpret_cast<const float*>(b), r, rowmap, n_parts, scaling,
                           reinterpret_cast<__bf16*>(et), ldet, N, RP);
    else if (b_dtype == FASTMAX_BF16)
        hipLaunchKernelGGL((lora_scatter_kernel<bf16_t>), grid, dim3(256), 0, s, reinterpret_cast<const bf16_t*>(b), r, rowmap, n_parts, scaling,
                           reinterpret_cast<__bf16*>(et), ldet, N, RP);
    else return FASTMAX_E_BAD_DTYPE;
    return (int)hipGetLastError();
}

int fastmax_hip_lora_scatter_backward(const void* d_et, int d_dtype, int64_t ldd, const int32_t* ind, const int32_t* part, float scaling,
                                      void* db, int b_dtype, int n_rows, int r, void* stream) {
    if (!d_et || !ind || !part || !db) return FASTMAX_E_NULL;
    if (n_rows <= 0 || r <= 0) return FASTMAX_E_BAD_SHAPE;
    const int64_t n = (int64_t)n_rows * r;
    const dim3 grid((unsigned)((n + 255) / 256));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define SCATTER_BWD(TB, TG)                                                                                                       \
    hipLaunchKernelGGL((lora_scatter_bwd_kernel<TB, TG>), grid, dim3(256), 0, s, reinterpret_cast<const TG*>(d_et), ldd, ind, part, scaling, \
                       reinterpret_cast<TB*>(db), n_rows, r)
    if (b_dtype == FASTMAX_F32 && d_dtype == FASTMAX_F32) SCATTER_BWD(float, float);
    else if (b_dtype == FASTMAX_F32 && d_dtype == FASTMAX_BF16) SCATTER_BWD(float, bf16_t);
    else if (b_dtype == FASTMAX_BF16 && d_dtype == FASTMAX_F32) SCATTER_BWD(bf16_t, float);
    else if (b_dtype == FASTMAX_BF16 && d_dtype == FASTMAX_BF16) SCATTER_BWD(bf16_t, bf16_t);
    else return FASTMAX_E_BAD_DTYPE;
#undef SCATTER_BWD
    return (int)hipGetLastError();
}

}  // extern "C"
