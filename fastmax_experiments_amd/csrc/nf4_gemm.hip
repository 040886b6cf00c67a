// QLoRA linear at training row counts: y[M][N] = x[M][K] . W[N][K]^T + bias + EA[M][RP] . EB[N][RP]^T   (gfx950)
//
// The frozen weight is either packed NF4 (decoded in the loop: lit_gpt/lora.py:170-177, 398-433 with bitsandbytes'
// Linear4bit underneath, finetune/lora.py:72-78) or already bf16 (a LoRALinear on a dense base, the lm-head).
// One workgroup of eight waves owns a 256 x 256 output tile and walks K in steps of 64:
//   * x tile (256 x 64 bf16 = 32 KB) comes in by LDS-DMA (global_load_lds_dwordx4), 16 bytes per lane, rows of 128 bytes with
//     the 16-byte chunk index XOR-ed with (row & 7) -- applied to the SOURCE address, the LDS side of a DMA is linear;
//   * W tile: a thread owns 32 codes (16 bytes) of one row, expands them through a 16-entry LDS table, scales by the block's
//     absmax (plain or double-quantised) and writes four 16-byte chunks of the same swizzled image.  With 256 rows of x per
//     decoded weight the decode costs ~1/4 of the matrix time instead of the ~1/2 of the 128-row kernel (nf4_lora.hip);
//   * two LDS stages (128 KB): tile t+1 is fetched / decoded while the 64 MFMAs of tile t run; one barrier per K step;
//   * wave (wm, wn) owns rows 128 wm.. and columns 64 wn..: 8 x 4 tiles of 16 x 16 in 128 accumulator registers.  The weight
//     is the MFMA A operand, x the B operand (both K-major: ds_read_b128 fragments);
//   * the LoRA branch is one more 32-deep step on the same accumulators; the tile leaves through LDS as whole 512-byte rows.
#include "fastmax_common.h"

#include <type_traits>

namespace fastmax {

typedef __bf16 gbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 gbf16x4 __attribute__((ext_vector_type(4)));
typedef float gf32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int gu32x4 __attribute__((ext_vector_type(4)));

struct GemmScale {            // block scales of the NF4 weight (see nf4_lora.hip)
    const float* absmax;
    const uint8_t* q;
    const float* absmax2;
    const float* code2;
    float offset;
    static __device__ __forceinline__ float mul_then_add(float a, float b, float c) {
#pragma clang fp contract(off)
        const float prod = a * b;
        return prod + c;
    }
    __device__ __forceinline__ float operator[](int64_t blk) const {
        return q ? mul_then_add(code2[q[blk]], absmax2[blk >> 8], offset) : absmax[blk];
    }
};

struct GemmParams {
    const __bf16* x;          // [M][K], leading dimension ldx
    const void* w;            // NF4: packed codes of W [N][K];  dense: bf16 W [N][K] (leading dimension K)
    GemmScale scale;
    const float* bias;        // [N] or null
    const __bf16* ea;         // [M][RP] or null
    const __bf16* eb;         // [N][RP] or null
    __bf16* y;                // [M][N], leading dimension ldy
    int M, N, K, RP;          // RP: 16 or 32 (rank padded)
    int64_t ldx, ldy;
    int nbn;                  // number of 256-column blocks
    int nbm, group_m, xcd;    // number of row blocks; row blocks per group of the workgroup -> tile map (0: none); XCD regrouping
    // RoPE + QKV de-interleave epilogue (qlora_gemm256a_kernel<.., true>): the product is the qkv projection of a batch of
    // sequences (rows = b * T + t; columns = (group, slot, d) of lit_gpt/model.py:397-420) and leaves as q (B, G qpk, T, hs),
    // k, v (B, G, T, hs) with the rotation of model.py:702-708 applied to the first rope_n elements of the q and k heads
    const float *rope_cos, *rope_sin;   // (T, rope_n) float32
    void *rq, *rk, *rv;
    int T, G, qpk, hs, rope_n, tables16;
};

__constant__ float kGemmNF4[16] = {-1.0f, -0.6961928009986877f, -0.5250730514526367f, -0.39491748809814453f,
                                   -0.28444138169288635f, -0.18477343022823334f, -0.09105003625154495f, 0.0f,
                                   0.07958029955625534f, 0.16093020141124725f, 0.24611230194568634f,
                                   0.33791524171829224f, 0.44070982933044434f, 0.5626170039176941f,
                                   0.7229568362236023f, 1.0f};

__device__ __forceinline__ int gsw(int row, int chunk) { return row * 128 + (((chunk ^ row) & 7) << 4); }

// workgroup id -> output tile.  Consecutive workgroup ids go round-robin over the 8 XCDs (each with its own 4 MB L2), so with
// `xcd` set the ids are first regrouped so that every XCD owns a CONTIGUOUS run of tile numbers; tiles are then numbered in groups
// of `group_m` row blocks, rows fastest: the ~32 tiles an XCD runs at a time form an 8 x 4 patch -- 12 operand tiles per K step
// to fetch into that L2 instead of 18 (column blocks fastest over 16 column blocks) or more.
__device__ __forceinline__ void gemm_tile_of(const GemmParams& prm, int& bm, int& bn) {
    int id = blockIdx.x;
    const int nwg = gridDim.x;
    if (prm.xcd && (nwg & 7) == 0) id = (id & 7) * (nwg >> 3) + (id >> 3);
    bn = id % prm.nbn;
    bm = id / prm.nbn;
    if (prm.group_m > 0) {
        const int per = prm.group_m * prm.nbn, grp = id / per, rem = id % per;
        const int rows = min(prm.group_m, prm.nbm - grp * prm.group_m);
        bm = grp * prm.group_m + rem % rows;
        bn = rem / rows;
    }
}

// host: the tile map of a launch.  "gemm_xcd" 1 (default): XCD regrouping + groups of 8 row blocks up to 16 column blocks; else
// column blocks fastest, in groups of "gemm_group_m" row blocks beyond 16 column blocks
static void gemm_map(GemmParams& p) {
    if (tune_get(TUNE_GEMM_XCD) && p.nbn <= 16) {                  // measured: +1.5 % (4096 wide) .. +3.5 % (2560 wide); -1 % at 48 column blocks
        p.xcd = 1;
        p.group_m = 8;
    } else {
        p.xcd = 0;
        p.group_m = (p.nbn > 16) ? tune_get(TUNE_GEMM_GROUP_M) : 0;
    }
}

namespace g256 {
constexpr int BM = 256, BN = 256, BK = 64;
constexpr int XT = BM * BK * 2, WT = BN * BK * 2, STAGE = XT + WT;          // 32 KB + 32 KB
constexpr int LUT = 2 * STAGE;                                               // 16 floats
constexpr int LDS_BYTES = LUT + 64;                                          // 131136
}  // namespace g256

// PIPE: the fragments of the next 32-deep half are read from LDS while the matrix instructions of the current half run (the
// default with the NF4 decode in the loop); PIPE = false is the plain loop -- every wave issues its copies, then reads, then
// multiplies -- kept as the A/B baseline ("gemm_sched" 14).  Round 2's other loop orders (partner-wave decode order, L2
// prefetch, copies between the MFMA groups, register staging, staggered copies, four-wave forms: profiles/r02_qlora_gemm.md)
// all measured slower and were removed in round 3.
template <bool WNF4, bool PIPE>
__global__ __launch_bounds__(512, 1) void qlora_gemm256_kernel(GemmParams prm) {
    using namespace g256;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* lut = reinterpret_cast<float*>(smem + LUT);
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 2, wn = w & 3;
    const int r = lane & 15, q4 = lane >> 4;
    // workgroup -> tile: column blocks fastest, so the workgroups that land on one XCD (ids congruent mod 8) share a few
    // column blocks of W (they stay in that XCD's L2) and stream over the rows of x
    // With many column blocks (the lm-head: 125) that order makes the 256 concurrent tiles two rows of blocks that stream ALL
    // of W per pass; in groups of `group_m` row blocks, rows fastest, the concurrent tiles form a 16 x 16 patch instead
    // (16 + 16 operand blocks per 256 tiles, not 2 + 125).
    int bn, bm;
    gemm_tile_of(prm, bm, bn);
    const int m0 = bm * BM, n0 = bn * BN;
    const int M = prm.M, N = prm.N, K = prm.K;
    if (WNF4 && tid < 16) lut[tid] = kGemmNF4[tid];

    // ---- x tile by LDS-DMA: instruction j of wave w fills rows 8 (4w + j) .. + 7 (1 KB); lane -> (row, slot) ------------
    const int drow = lane >> 3, dslot = lane & 7;
    const int dchunk = dslot ^ drow;                                 // the row's low three bits are drow (rows come in 8s)
    auto dma_tile = [&](const __bf16* base, int64_t ld, int row0, int nrows, int k0, char* dst) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = 8 * (4 * w + j) + drow;
            const int gr = min(row0 + row, nrows - 1);               // rows past the end re-read the last row (never stored)
            const __bf16* src = base + (int64_t)gr * ld + k0 + 8 * dchunk;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(dst + (4 * w + j) * 1024), 16, 0, 0);
        }
    };
    // ---- W tile from NF4 codes: thread -> row wrow, 32 codes of half whalf ----------------------------------------------
    const int wrow = tid >> 1, whalf = tid & 1;
    const int wn_g = min(n0 + wrow, N - 1);
    gu32x4 wr;
    float wa;
    auto load_codes = [&](int k0) {
        const int64_t e = (int64_t)wn_g * K + k0 + 32 * whalf;
        wr = *reinterpret_cast<const gu32x4*>(reinterpret_cast<const uint8_t*>(prm.w) + (e >> 1));
        wa = prm.scale[e >> 6];
    };
    // words [w0, w1) of the thread's four code words -> bf16 -> the W image (one 16-byte chunk per word)
    auto decode_words = [&](char* dst, int w0, int w1) {
#pragma unroll
        for (int wd = w0; wd < w1; ++wd) {
            const unsigned int v = wr[wd];
            gbf16x8 o;
#pragma unroll
            for (int by = 0; by < 4; ++by) {
                const unsigned int byte = (v >> (8 * by)) & 0xffu;
                o[2 * by] = (__bf16)(lut[byte >> 4] * wa);           // high nibble first
                o[2 * by + 1] = (__bf16)(lut[byte & 15u] * wa);
            }
            *reinterpret_cast<gbf16x8*>(dst + gsw(wrow, 4 * whalf + wd)) = o;
        }
    };
    auto decode_codes = [&](char* dst) { decode_words(dst, 0, 4); };
    gf32x4 acc[4][8];                                                // [nt][mt]: rows n (registers), column m (lane)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = gf32x4{0, 0, 0, 0};

    auto mma_k32 = [&](const char* Xs, const char* Ws, int ks) {
        gbf16x8 af[4], bfm[8];
#pragma unroll
        for (int t = 0; t < 4; ++t) af[t] = *reinterpret_cast<const gbf16x8*>(Ws + gsw(64 * wn + 16 * t + r, 4 * ks + q4));
#pragma unroll
        for (int t = 0; t < 8; ++t) bfm[t] = *reinterpret_cast<const gbf16x8*>(Xs + gsw(128 * wm + 16 * t + r, 4 * ks + q4));
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int mt = 0; mt < 8; ++mt)
                acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[nt], bfm[mt], acc[nt][mt], 0, 0, 0);
    };

    // PIPE: the fragments of the next 32-deep half are read from LDS while the matrix instructions of the current half run.
    // With "read 12 fragments, then 32 MFMAs" every wave leaves the step's barrier at the same moment, so all eight read LDS
    // together and then all multiply together: neither unit overlaps the other (~3500 cycles per step against 2048 of MFMA).
    auto read_frags = [&](const char* Xs, const char* Ws, int ks, gbf16x8 (&af)[4], gbf16x8 (&bfm)[8]) {
#pragma unroll
        for (int t = 0; t < 4; ++t) af[t] = *reinterpret_cast<const gbf16x8*>(Ws + gsw(64 * wn + 16 * t + r, 4 * ks + q4));
#pragma unroll
        for (int t = 0; t < 8; ++t) bfm[t] = *reinterpret_cast<const gbf16x8*>(Xs + gsw(128 * wm + 16 * t + r, 4 * ks + q4));
    };
    const int KT = K / BK;
    // prologue: tile 0 -> stage 0; the codes of tile 1 wait in registers
    dma_tile(prm.x, prm.ldx, m0, M, 0, smem);
    if constexpr (WNF4) {
        load_codes(0);
        __syncthreads();                                             // lut visible
        decode_codes(smem + XT);
        if (KT > 1) load_codes(BK);
    } else {
        dma_tile(reinterpret_cast<const __bf16*>(prm.w), K, n0, N, 0, smem + XT);
    }
    __syncthreads();
    if constexpr (PIPE) {
        // fragments: the four weight fragments of a half are double-buffered (a0 / a1); the eight x fragments rotate in place --
        // the loop runs row tile by row tile, and as soon as the four MFMAs of row tile mt have been issued b[mt] is refilled
        // with the next half's fragment
        gbf16x8 a0[4], a1[4], b[8];
        auto half = [&](const gbf16x8 (&af)[4], gbf16x8 (&afn)[4], const char* nXs, const char* nWs, int nks, bool fetch) {
            if (fetch) {
#pragma unroll
                for (int t = 0; t < 4; ++t) afn[t] = *reinterpret_cast<const gbf16x8*>(nWs + gsw(64 * wn + 16 * t + r, 4 * nks + q4));
            }
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[nt], b[mt], acc[nt][mt], 0, 0, 0);
                if (fetch) b[mt] = *reinterpret_cast<const gbf16x8*>(nXs + gsw(128 * wm + 16 * mt + r, 4 * nks + q4));
            }
        };
        if (KT > 1) {
            dma_tile(prm.x, prm.ldx, m0, M, BK, smem + STAGE);
            if constexpr (!WNF4) dma_tile(reinterpret_cast<const __bf16*>(prm.w), K, n0, N, BK, smem + STAGE + XT);
        }
        read_frags(smem, smem + XT, 0, a0, b);
        for (int kt = 0; kt < KT; ++kt) {
            char* cur = smem + (kt & 1) * STAGE;
            char* nxt = smem + ((kt & 1) ^ 1) * STAGE;
            const bool more = kt + 1 < KT;
            half(a0, a1, cur, cur + XT, 1, true);
            if constexpr (WNF4) {
                if (more) decode_codes(nxt + XT);                    // nxt: last read (tile kt-1) before the previous barrier
            }
            __syncthreads();                                         // tile kt+1 whole in nxt; nobody reads cur any more
            if (kt + 2 < KT) {
                dma_tile(prm.x, prm.ldx, m0, M, (kt + 2) * BK, cur);
                if constexpr (WNF4) load_codes((kt + 2) * BK);
                else dma_tile(reinterpret_cast<const __bf16*>(prm.w), K, n0, N, (kt + 2) * BK, cur + XT);
            }
            const char* nf = more ? nxt : cur;                       // last step: a harmless re-read instead of a branch per tile
            half(a1, a0, nf, nf + XT, 0, true);
        }
        __syncthreads();
    } else
    for (int kt = 0; kt < KT; ++kt) {
        char* cur = smem + (kt & 1) * STAGE;
        char* nxt = smem + ((kt & 1) ^ 1) * STAGE;
        const bool more = kt + 1 < KT;
        if (more) {
            dma_tile(prm.x, prm.ldx, m0, M, (kt + 1) * BK, nxt);
            if constexpr (!WNF4) dma_tile(reinterpret_cast<const __bf16*>(prm.w), K, n0, N, (kt + 1) * BK, nxt + XT);
        }
        mma_k32(cur, cur + XT, 0);
        mma_k32(cur, cur + XT, 1);
        if constexpr (WNF4) {                                        // every wave decodes tile kt+1 after its matrix instructions
            if (more) decode_codes(nxt + XT);
            if (kt + 2 < KT) load_codes((kt + 2) * BK);
        }
        __syncthreads();                                             // tile kt+1 landed (DMA drained by the barrier's wait), tile kt consumed
    }
    // ---- LoRA branch: one more step over the padded rank ------------------------------------------------------------------
    if (prm.ea && prm.eb) {
        // EA rows m0.. -> X image, EB rows n0.. -> W image (stage 0); RP = 16 or 32 columns = 2 or 4 chunks per row
        const int cpr = prm.RP / 8;
        for (int i = tid; i < BM * cpr; i += 512) {
            const int row = i / cpr, c = i % cpr;
            const int gm = min(m0 + row, M - 1), gn = min(n0 + row, N - 1);
            *reinterpret_cast<gbf16x8*>(smem + gsw(row, c)) = *reinterpret_cast<const gbf16x8*>(prm.ea + (int64_t)gm * prm.RP + 8 * c);
            *reinterpret_cast<gbf16x8*>(smem + XT + gsw(row, c)) = *reinterpret_cast<const gbf16x8*>(prm.eb + (int64_t)gn * prm.RP + 8 * c);
        }
        if (prm.RP == 16) {                                          // zero the upper half of the 32-deep step
            for (int i = tid; i < BM * 2; i += 512) {
                const int row = i >> 1, c = 2 + (i & 1);
                const gbf16x8 z = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
                *reinterpret_cast<gbf16x8*>(smem + gsw(row, c)) = z;
                *reinterpret_cast<gbf16x8*>(smem + XT + gsw(row, c)) = z;
            }
        }
        __syncthreads();
        mma_k32(smem, smem + XT, 0);
        __syncthreads();
    }
    // ---- epilogue: tile -> LDS as [256 m][256 n] bf16 (512-byte rows, 16-byte chunk index XOR-ed with m & 31), then rows out ---
    char* ct = smem;
    gf32x4 bias4[4];                                                 // the lane's four columns of every column tile, fetched once
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int gn = min(n0 + 64 * wn + 16 * nt + 4 * q4, N - 4);
        bias4[nt] = prm.bias ? *reinterpret_cast<const gf32x4*>(prm.bias + gn) : gf32x4{0, 0, 0, 0};
    }
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
        const int m = 128 * wm + 16 * mt + r;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int n = 64 * wn + 16 * nt + 4 * q4;                // 4 consecutive columns = 8 bytes
            const gf32x4 v = acc[nt][mt] + bias4[nt];
            gbf16x4 o;
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = (__bf16)v[i];
            const int chunk = n >> 3;                                // 32 chunks per row
            *reinterpret_cast<gbf16x4*>(ct + m * 512 + (((chunk ^ m) & 31) << 4) + ((n & 4) << 1)) = o;
        }
    }
    __syncthreads();
    for (int i = tid; i < BM * 32; i += 512) {
        const int m = i >> 5, chunk = i & 31;
        const int gm = m0 + m, gn = n0 + 8 * chunk;
        if (gm < M && gn < N) {
            const gu32x4 v = *reinterpret_cast<const gu32x4*>(ct + m * 512 + (((chunk ^ m) & 31) << 4));
            *reinterpret_cast<gu32x4*>(prm.y + (int64_t)gm * prm.ldy + gn) = v;
        }
    }
}

// ---- experiment ("gemm_sched" 16): the default loop with an UNEVEN row split inside the SIMD pairs.  In the default loop the
// waves that issue the copies (w < 4) are the critical path of a step: 16 LDS-DMA issues (~80 cycles each) + their 64 MFMAs,
// while their partners only have 64 MFMAs.  Here the copy-issuing wave of a pair owns 7 of its column strip's 16 row tiles and
// the partner 9 (56 / 72 MFMAs per step).  Dense bf16 weight, plain epilogue (bias + LoRA step).  The body is instantiated
// twice (7 and 9 row tiles) behind a wave-uniform branch, so every loop is fully unrolled without per-tile predicates.
// x cos + y sin with the reference's three float32 roundings (no fused multiply-add), as fastmax_rope.hip
__device__ __forceinline__ float gemm_rope_mul_add(float x, float c, float y, float s_) {
#pragma clang fp contract(off)
    const float p0 = x * c;
    const float p1 = y * s_;
    return p0 + p1;
}

template <int MTA, bool ROPE = false>   // MTA: row tiles (of 16) of the copy-issuing wave; its partner owns 16 - MTA
__global__ __launch_bounds__(512, 1) void qlora_gemm256a_kernel(GemmParams prm) {
    using namespace g256;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 2, wn = w & 3;
    const int r = lane & 15, q4 = lane >> 4;
    int bn, bm;
    gemm_tile_of(prm, bm, bn);
    const int m0 = bm * BM, n0 = bn * BN;
    const int M = prm.M, N = prm.N, K = prm.K;
    const int drow = lane >> 3, dslot = lane & 7, dchunk = dslot ^ drow;
    const int KT = K / BK;
    // copies: waves 0-3 only, eight 1 KB pieces of each operand tile per wave
    auto dma_tile = [&](const __bf16* base, int64_t ld, int row0, int nrows, int k0, char* dst) {
        if (w < 4) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int pc = 8 * w + j, row = 8 * pc + drow;
                const int gr = min(row0 + row, nrows - 1);
                const __bf16* src = base + (int64_t)gr * ld + k0 + 8 * dchunk;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(dst + pc * 1024), 16, 0, 0);
            }
        }
    };
    const __bf16* wbase = reinterpret_cast<const __bf16*>(prm.w);
    dma_tile(prm.x, prm.ldx, m0, M, 0, smem);
    dma_tile(wbase, K, n0, N, 0, smem + XT);
    __syncthreads();
    if (KT > 1) {
        dma_tile(prm.x, prm.ldx, m0, M, BK, smem + STAGE);
        dma_tile(wbase, K, n0, N, BK, smem + STAGE + XT);
    }
    char* ct = smem;                                                 // epilogue image [256 m][256 n] bf16
    auto body = [&](auto mt_tag) {
        constexpr int MT = decltype(mt_tag)::value;
        const int mrow0 = MT == MTA ? 0 : 16 * MTA;                  // rows 0 .. 16 MTA - 1 / the rest
        gf32x4 acc[4][MT];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < MT; ++j) acc[i][j] = gf32x4{0, 0, 0, 0};
        gbf16x8 a0[4], a1[4], b[MT];
        auto frag_a = [&](const char* Ws, int ks, int t) { return *reinterpret_cast<const gbf16x8*>(Ws + gsw(64 * wn + 16 * t + r, 4 * ks + q4)); };
        auto frag_b = [&](const char* Xs, int ks, int t) { return *reinterpret_cast<const gbf16x8*>(Xs + gsw(mrow0 + 16 * t + r, 4 * ks + q4)); };
        auto half = [&](const gbf16x8 (&af)[4], gbf16x8 (&afn)[4], const char* nst, int nks) {
#pragma unroll
            for (int t = 0; t < 4; ++t) afn[t] = frag_a(nst + XT, nks, t);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[nt], b[mt], acc[nt][mt], 0, 0, 0);
                b[mt] = frag_b(nst, nks, mt);
            }
        };
#pragma unroll
        for (int t = 0; t < 4; ++t) a0[t] = frag_a(smem + XT, 0, t);
#pragma unroll
        for (int t = 0; t < MT; ++t) b[t] = frag_b(smem, 0, t);
        for (int kt = 0; kt < KT; ++kt) {
            char* cur = smem + (kt & 1) * STAGE;
            char* nxt = smem + ((kt & 1) ^ 1) * STAGE;
            half(a0, a1, cur, 1);
            __syncthreads();                                         // tile kt+1 whole in nxt; nobody reads cur any more
            if (kt + 2 < KT) {
                dma_tile(prm.x, prm.ldx, m0, M, (kt + 2) * BK, cur);
                dma_tile(wbase, K, n0, N, (kt + 2) * BK, cur + XT);
            }
            half(a1, a0, (kt + 1 < KT) ? nxt : cur, 0);              // last step: a harmless re-read instead of a branch
        }
        __syncthreads();
        // LoRA branch: one more 32-deep step over the padded rank
        if (prm.ea && prm.eb) {
            const int cpr = prm.RP / 8;
            const gbf16x8 z = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
            for (int i = tid; i < BM * 4; i += 512) {
                const int row = i >> 2, c = i & 3;
                const int gm = min(m0 + row, M - 1), gn = min(n0 + row, N - 1);
                *reinterpret_cast<gbf16x8*>(smem + gsw(row, c)) = c < cpr ? *reinterpret_cast<const gbf16x8*>(prm.ea + (int64_t)gm * prm.RP + 8 * c) : z;
                *reinterpret_cast<gbf16x8*>(smem + XT + gsw(row, c)) = c < cpr ? *reinterpret_cast<const gbf16x8*>(prm.eb + (int64_t)gn * prm.RP + 8 * c) : z;
            }
            __syncthreads();
#pragma unroll
            for (int t = 0; t < 4; ++t) a0[t] = frag_a(smem + XT, 0, t);
#pragma unroll
            for (int t = 0; t < MT; ++t) b[t] = frag_b(smem, 0, t);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[nt], b[mt], acc[nt][mt], 0, 0, 0);
            __syncthreads();
        }
        // tile -> LDS as [256 m][256 n] bf16 (512-byte rows, 16-byte chunk index XOR-ed with m & 31)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int gn = min(n0 + 64 * wn + 16 * nt + 4 * q4, N - 4);
            const gf32x4 bias4 = prm.bias ? *reinterpret_cast<const gf32x4*>(prm.bias + gn) : gf32x4{0, 0, 0, 0};
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int m = mrow0 + 16 * mt + r, n = 64 * wn + 16 * nt + 4 * q4;
                const gf32x4 v = acc[nt][mt] + bias4;
                gbf16x4 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = (__bf16)v[i];
                const int chunk = n >> 3;
                *reinterpret_cast<gbf16x4*>(ct + m * 512 + (((chunk ^ m) & 31) << 4) + ((n & 4) << 1)) = o;
            }
        }
    };
    if (wm == 0) body(std::integral_constant<int, MTA>{});
    else body(std::integral_constant<int, 16 - MTA>{});
    __syncthreads();
    if constexpr (ROPE) {
        // the tile holds, per token row, whole heads of the qkv projection (256 % hs == 0): every 16-byte piece goes to its
        // (q | k | v, head, token) row; pieces of the rotated range of a q or k head are combined with their partner piece
        // (d <-> d +- rope_n / 2) from the same LDS row first -- the values are the bf16-rounded outputs, so the result is
        // bit-identical to the separate pass of fastmax_rope.hip over a stored qkv tensor
        // thread -> (token row, 16-byte offset d inside a head); it visits the 256 / hs heads of the tile that share this row:
        // the rotation's cos / sin pieces depend on (token, d) only, so they are loaded ONCE per row and serve every head
        // (fetched per piece they are four times the bytes of the data).  d, rotate / partner and the per-head destinations are
        // thread constants (the row advances by 512 / (hs / 8) per visit).
        const int hs = prm.hs, total = prm.qpk + 2, half = prm.rope_n >> 1, H = prm.G * prm.qpk;
        const int cph = hs >> 3, nh = 256 / hs;                       // 16-byte chunks per head; heads per tile row (2 or 4 or 8)
        const int j = tid % cph, d = 8 * j;
        const bool in_rope = d < prm.rope_n, first = d < half;
        const int pj = first ? j + (half >> 3) : j - (half >> 3);
        for (int m = tid / cph; m < BM && m0 + m < M; m += 512 / cph) {
            const int gm = m0 + m;
            const int b = gm / prm.T, t = gm - b * prm.T;
            gf32x4 c0 = {0, 0, 0, 0}, c1 = c0, s0 = c0, s1 = c0;
            if (in_rope) {
                const gf32x4* cs = reinterpret_cast<const gf32x4*>(prm.rope_cos + (int64_t)t * prm.rope_n + d);
                const gf32x4* sn = reinterpret_cast<const gf32x4*>(prm.rope_sin + (int64_t)t * prm.rope_n + d);
                c0 = cs[0]; c1 = cs[1]; s0 = sn[0]; s1 = sn[1];
            }
            for (int hh = 0; hh < nh; ++hh) {
                const int f = n0 + hh * hs + d;
                if (f >= N) break;
                const int hd = f / hs;                                // = n0 / hs + hh
                const int g = hd / total, slot = hd - g * total;
                const int chunk = hh * cph + j;
                gbf16x8 xo = *reinterpret_cast<const gbf16x8*>(ct + m * 512 + (((chunk ^ m) & 31) << 4));
                if (in_rope && slot <= prm.qpk) {
                    const int pchunk = hh * cph + pj;
                    const gbf16x8 xp = *reinterpret_cast<const gbf16x8*>(ct + m * 512 + (((pchunk ^ m) & 31) << 4));
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float a = (float)xo[e], pb = first ? -(float)xp[e] : (float)xp[e];
                        const float c = e < 4 ? c0[e & 3] : c1[e & 3], s_ = e < 4 ? s0[e & 3] : s1[e & 3];
                        // first half: x[d] cos[d] - x[d + half] sin[d];  second: x[d] cos[d] + x[d - half] sin[d]   (model.py:702-708)
                        const float o = prm.tables16 ? (float)(__bf16)(a * c) + (float)(__bf16)(pb * s_) : gemm_rope_mul_add(a, c, pb, s_);
                        xo[e] = (__bf16)o;
                    }
                }
                __bf16* dst = slot < prm.qpk ? reinterpret_cast<__bf16*>(prm.rq) + (((int64_t)b * H + g * prm.qpk + slot) * prm.T + t) * hs
                            : (slot == prm.qpk ? reinterpret_cast<__bf16*>(prm.rk) : reinterpret_cast<__bf16*>(prm.rv)) +
                                  (((int64_t)b * prm.G + g) * prm.T + t) * hs;
                *reinterpret_cast<gbf16x8*>(dst + d) = xo;
            }
        }
        return;
    }
    for (int i = tid; i < BM * 32; i += 512) {
        const int m = i >> 5, chunk = i & 31;
        const int gm = m0 + m, gn = n0 + 8 * chunk;
        if (gm < M && gn < N) {
            const gu32x4 v = *reinterpret_cast<const gu32x4*>(ct + m * 512 + (((chunk ^ m) & 31) << 4));
            *reinterpret_cast<gu32x4*>(prm.y + (int64_t)gm * prm.ldy + gn) = v;
        }
    }
}

template <int MTA, bool ROPE = false>
static int launch_gemm256a(GemmParams p, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(qlora_gemm256a_kernel<MTA, ROPE>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           g256::LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    p.nbm = (p.M + 255) / 256;
    gemm_map(p);
    hipLaunchKernelGGL((qlora_gemm256a_kernel<MTA, ROPE>), dim3(p.nbn * p.nbm), dim3(512), g256::LDS_BYTES, stream, p);
    return (int)hipGetLastError();
}

// The same product for row counts whose 256 x 256 tiles would leave CUs idle (M = 2048 .. 8192 at the layer widths of the
// fine-tune configs: 64 .. 160 tiles on 256 CUs): 128 x 256 output tiles, dense bf16 weight (the decode-once route), the same
// images, loop and epilogue.  Eight waves as 2 (m) x 4 (n): a wave owns 64 x 64 (16 accumulator tiles); per K step a wave
// issues 2 + 4 LDS-DMA pieces and 32 MFMAs.  Two stages of 16 KB (x) + 32 KB (W) = 96 KB.
namespace g128 {
constexpr int BM = 128, BN = 256, BK = 64;
constexpr int XT = BM * BK * 2, WT = BN * BK * 2, STAGE = XT + WT;
constexpr int LDS_BYTES = 2 * STAGE;                                         // 98304
}  // namespace g128

__global__ __launch_bounds__(512, 1) void qlora_gemm128_kernel(GemmParams prm) {
    using namespace g128;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 2, wn = w & 3;
    const int r = lane & 15, q4 = lane >> 4;
    int bn, bm;
    gemm_tile_of(prm, bm, bn);
    const int m0 = bm * BM, n0 = bn * BN;
    const int M = prm.M, N = prm.N, K = prm.K;
    const int drow = lane >> 3, dslot = lane & 7, dchunk = dslot ^ drow;
    // piece pc of an operand tile = its rows 8 pc .. 8 pc + 7 (1 KB)
    auto dma_piece = [&](const __bf16* base, int64_t ld, int row0, int nrows, int k0, char* dst, int pc) {
        const int gr = min(row0 + 8 * pc + drow, nrows - 1);         // rows past the end re-read the last row (never stored)
        const __bf16* src = base + (int64_t)gr * ld + k0 + 8 * dchunk;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(dst + pc * 1024), 16, 0, 0);
    };
    // copies issued by one wave of each SIMD pair (w < 4; w and w + 4 share a SIMD): its partner goes straight to the matrix
    // instructions (the measured winner of the 256-row kernel's loop orders)
    auto dma_tiles = [&](int k0, char* stage) {
        if (w < 4) {
#pragma unroll
            for (int j = 0; j < 4; ++j) dma_piece(prm.x, prm.ldx, m0, M, k0, stage, 4 * w + j);
#pragma unroll
            for (int j = 0; j < 8; ++j) dma_piece(reinterpret_cast<const __bf16*>(prm.w), K, n0, N, k0, stage + XT, 8 * w + j);
        }
    };
    gf32x4 acc[4][4];                                                // [nt][mt]: rows n (registers), column m (lane)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = gf32x4{0, 0, 0, 0};
    auto mma_k32 = [&](const char* Xs, const char* Ws, int ks) {
        gbf16x8 af[4], bfm[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) af[t] = *reinterpret_cast<const gbf16x8*>(Ws + gsw(64 * wn + 16 * t + r, 4 * ks + q4));
#pragma unroll
        for (int t = 0; t < 4; ++t) bfm[t] = *reinterpret_cast<const gbf16x8*>(Xs + gsw(64 * wm + 16 * t + r, 4 * ks + q4));
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
                acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[nt], bfm[mt], acc[nt][mt], 0, 0, 0);
    };
    const int KT = K / BK;
    dma_tiles(0, smem);
    __syncthreads();
    for (int kt = 0; kt < KT; ++kt) {
        char* cur = smem + (kt & 1) * STAGE;
        char* nxt = smem + ((kt & 1) ^ 1) * STAGE;
        if (kt + 1 < KT) dma_tiles((kt + 1) * BK, nxt);
        mma_k32(cur, cur + XT, 0);
        mma_k32(cur, cur + XT, 1);
        __syncthreads();                                             // tile kt+1 landed, tile kt consumed
    }
    // ---- LoRA branch: one more 32-deep step over the padded rank (EA rows m0.. -> X image, EB rows n0.. -> W image) -------
    if (prm.ea && prm.eb) {
        const int cpr = prm.RP / 8;
        const gbf16x8 z = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
        for (int i = tid; i < BN * 4; i += 512) {
            const int row = i >> 2, c = i & 3;
            const int gn = min(n0 + row, N - 1);
            *reinterpret_cast<gbf16x8*>(smem + XT + gsw(row, c)) =
                c < cpr ? *reinterpret_cast<const gbf16x8*>(prm.eb + (int64_t)gn * prm.RP + 8 * c) : z;
            if (row < BM) {
                const int gm = min(m0 + row, M - 1);
                *reinterpret_cast<gbf16x8*>(smem + gsw(row, c)) =
                    c < cpr ? *reinterpret_cast<const gbf16x8*>(prm.ea + (int64_t)gm * prm.RP + 8 * c) : z;
            }
        }
        __syncthreads();
        mma_k32(smem, smem + XT, 0);
        __syncthreads();
    }
    // ---- epilogue: tile -> LDS as [128 m][256 n] bf16 (512-byte rows, 16-byte chunk index XOR-ed with m & 31), then rows out ---
    char* ct = smem;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int gn = min(n0 + 64 * wn + 16 * nt + 4 * q4, N - 4);
        const gf32x4 bias4 = prm.bias ? *reinterpret_cast<const gf32x4*>(prm.bias + gn) : gf32x4{0, 0, 0, 0};
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int m = 64 * wm + 16 * mt + r, n = 64 * wn + 16 * nt + 4 * q4;
            const gf32x4 v = acc[nt][mt] + bias4;
            gbf16x4 o;
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = (__bf16)v[i];
            const int chunk = n >> 3;
            *reinterpret_cast<gbf16x4*>(ct + m * 512 + (((chunk ^ m) & 31) << 4) + ((n & 4) << 1)) = o;
        }
    }
    __syncthreads();
    for (int i = tid; i < BM * 32; i += 512) {
        const int m = i >> 5, chunk = i & 31;
        const int gm = m0 + m, gn = n0 + 8 * chunk;
        if (gm < M && gn < N) {
            const gu32x4 v = *reinterpret_cast<const gu32x4*>(ct + m * 512 + (((chunk ^ m) & 31) << 4));
            *reinterpret_cast<gu32x4*>(prm.y + (int64_t)gm * prm.ldy + gn) = v;
        }
    }
}

static int launch_gemm128(GemmParams p, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(qlora_gemm128_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           g128::LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    p.nbm = (p.M + 127) / 128;
    gemm_map(p);
    hipLaunchKernelGGL(qlora_gemm128_kernel, dim3(p.nbn * p.nbm), dim3(512), g128::LDS_BYTES, stream, p);
    return (int)hipGetLastError();
}

// W^T as a dense bf16 matrix [K][N] from the NF4 codes of W [N][K]: the operand of dx = dy . W through the same GEMM kernel
// (contraction over n needs the weight n-major).  One workgroup = a 64 (n) x 64 (k) tile: 32 bytes of codes per row (one
// 64-weight block, one scale), decoded into LDS, written out as 64 rows of 128 bytes.  grid = (K/64, N/64), block = 256.
__global__ __launch_bounds__(256) void nf4_dequant_transposed_kernel(const uint8_t* wq, GemmScale scale, __bf16* out, int N, int K) {
    __shared__ float lut[16];
    __shared__ __bf16 tile[64][66];                                  // [k][n], padded
    const int tid = threadIdx.x;
    if (tid < 16) lut[tid] = kGemmNF4[tid];
    __syncthreads();
    const int k0 = blockIdx.x * 64, n0 = blockIdx.y * 64;
    const int n = tid >> 2, part = tid & 3;                          // 4 threads per row: 8 bytes = 16 codes each
    const int64_t e = (int64_t)(n0 + n) * K + k0 + 16 * part;
    const unsigned long long pk = *reinterpret_cast<const unsigned long long*>(wq + (e >> 1));
    const float a = scale[e >> 6];
#pragma unroll
    for (int by = 0; by < 8; ++by) {
        const unsigned int byte = (unsigned int)(pk >> (8 * by)) & 0xffu;
        tile[16 * part + 2 * by][n] = (__bf16)(lut[byte >> 4] * a);
        tile[16 * part + 2 * by + 1][n] = (__bf16)(lut[byte & 15u] * a);
    }
    __syncthreads();
    const int kr = tid >> 2, seg = tid & 3;                          // row k0 + kr, 16 columns (32 bytes) per thread
    __bf16* dst = out + (int64_t)(k0 + kr) * N + n0 + 16 * seg;
#pragma unroll
    for (int c = 0; c < 16; c += 2) {
        const unsigned int lo = __builtin_bit_cast(unsigned short, tile[kr][16 * seg + c]);
        const unsigned int hi = __builtin_bit_cast(unsigned short, tile[kr][16 * seg + c + 1]);
        reinterpret_cast<unsigned int*>(dst)[c >> 1] = lo | (hi << 16);
    }
}

template <bool WNF4, bool PIPE>
static int launch_gemm256(GemmParams p, hipStream_t stream) {
    auto kern = qlora_gemm256_kernel<WNF4, PIPE>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, g256::LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const int nbm = (p.M + 255) / 256;
    p.nbm = nbm;
    gemm_map(p);
    hipLaunchKernelGGL(kern, dim3(p.nbn * nbm), dim3(512), g256::LDS_BYTES, stream, p);
    return (int)hipGetLastError();
}

}  // namespace fastmax

using namespace fastmax;

extern "C" {

// y = x W^T (+ bias) (+ ea eb^T): W as NF4 codes with `scales` (w_is_nf4 != 0) or as a dense bf16 matrix (scales ignored).
// bf16 activations; needs K % 64 == 0, N % 8 == 0, 16-byte aligned rows; rank_pad 16 or 32 when ea / eb are given.
int fastmax_hip_qlora_gemm(const void* x, int64_t ldx, const void* w, int w_is_nf4, const fastmax_nf4_scales* scales,
                           const float* bias, const void* ea, const void* eb, int rank_pad, void* y, int64_t ldy, int M, int N,
                           int K, void* stream) {
    if (!x || !w || !y) return FASTMAX_E_NULL;
    if (M <= 0 || N <= 0 || K <= 0 || (K % 64) || (N % 8)) return FASTMAX_E_BAD_SHAPE;
    if ((ea == nullptr) != (eb == nullptr)) return FASTMAX_E_NULL;
    if (ea && rank_pad != 16 && rank_pad != 32) return FASTMAX_E_BAD_SHAPE;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(w)) & 15) return FASTMAX_E_ALIGNMENT;
    if (((ldx * 2) & 15) || ((ldy * 2) & 15)) return FASTMAX_E_ALIGNMENT;
    if (bias && (reinterpret_cast<uintptr_t>(bias) & 15)) return FASTMAX_E_ALIGNMENT;
    if (ea && ((reinterpret_cast<uintptr_t>(ea) | reinterpret_cast<uintptr_t>(eb)) & 15)) return FASTMAX_E_ALIGNMENT;
    GemmScale sc{nullptr, nullptr, nullptr, nullptr, 0.f};
    if (w_is_nf4) {
        if (!scales) return FASTMAX_E_NULL;
        if (scales->absmax_q ? !(scales->absmax2 && scales->code2) : !scales->absmax) return FASTMAX_E_NULL;
        sc = GemmScale{scales->absmax, scales->absmax_q, scales->absmax2, scales->code2, scales->offset};
    }
    GemmParams p{reinterpret_cast<const __bf16*>(x), w, sc, bias, reinterpret_cast<const __bf16*>(ea),
                 reinterpret_cast<const __bf16*>(eb), reinterpret_cast<__bf16*>(y), M, N, K, rank_pad, ldx, ldy, (N + 255) / 256, 0, 0, 0,
                 nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0, 0, 0};
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    // Tuning key "gemm_sched" (A/B): 0 = defaults below; 14 = the plain loop (every wave issues its copies, reads its fragments,
    // multiplies) for either weight form; 12 / 13 force / forbid the 128-row tiles.  The other loop orders of round 2 are gone.
    const int sched = tune_get(TUNE_GEMM_SCHED);
    if (!w_is_nf4) {
        // 256 x 256 tiles for at most half of the 256 CUs: 128-row tiles fill the chip in one round (measured: (4096, 2048, 2048)
        // 0.054 -> 0.044 ms, (2048, 2560, 2048) 0.052 -> 0.035; with 160 tiles, e.g. (4096, 2560, 2048), the 256-row tiles stay
        // ahead, 0.055 vs 0.068: two rounds of half tiles cost more than one round of whole ones)
        if (sched != 13 && (sched == 12 || (int64_t)((M + 255) / 256) * ((N + 255) / 256) <= 128)) return launch_gemm128(p, st);
        if (sched == 14) return launch_gemm256<false, false>(p, st);
        // default: copies issued by one wave of each SIMD pair, fragments of the next half read under the MFMAs of this one
        // (5-9 % ahead of the plain loop at every fine-tune shape), the copy-issuing wave of a pair owning 7 of the strip's 16
        // row tiles and its partner 9 (another 2-3 %) -- profiles/r02_qlora_gemm.md; same bits as the plain loop
        return launch_gemm256a<7>(p, st);
    }
    // NF4 decoded in the loop (FASTMAX_QLORA_ROUTE=fused; the default route decodes once and takes the dense kernels above):
    // fragments of the next half read under the MFMAs, every wave issues its own x copies
    if (sched == 14) return launch_gemm256<true, false>(p, st);
    return launch_gemm256<true, true>(p, st);
}

// The qkv projection of an attention sub-layer with its neighbours fused into the tile's way out (SURVEY.md 8f row 1 inside
// the ★ row): y = x W^T + bias + ea eb^T is never stored as (tokens, qkv features); every 16-byte piece of the finished tile
// goes to q (B, G q_per_kv, T, hs) / k, v (B, G, T, hs), the first rope_n elements of the q and k heads rotated
// (lit_gpt/model.py:397-425, 702-708).  x rows are b * T + t.  Dense bf16 weight [N][K], N == G (q_per_kv + 2) hs,
// 256 % hs == 0, rope_n % 16 == 0, K % 64 == 0; tables16: the caller's rope cache was 16-bit (products rounded before the sum).
int fastmax_hip_qlora_gemm_rope(const void* x, int64_t ldx, const void* w, const float* bias, const void* ea, const void* eb,
                                int rank_pad, const float* cos, const float* sin, void* q, void* k, void* v, int M, int N, int K,
                                int T, int G, int q_per_kv, int head_size, int rope_n_elem, int tables16, void* stream) {
    if (!x || !w || !cos || !sin || !q || !k || !v) return FASTMAX_E_NULL;
    if (M <= 0 || N <= 0 || K <= 0 || (K % 64) || T <= 0 || (M % T) || G <= 0 || q_per_kv <= 0 || head_size <= 0) return FASTMAX_E_BAD_SHAPE;
    if (N != G * (q_per_kv + 2) * head_size || (256 % head_size) || rope_n_elem < 0 || rope_n_elem > head_size || (rope_n_elem % 16))
        return FASTMAX_E_BAD_SHAPE;
    if ((ea == nullptr) != (eb == nullptr)) return FASTMAX_E_NULL;
    if (ea && rank_pad != 16 && rank_pad != 32) return FASTMAX_E_BAD_SHAPE;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k) |
         reinterpret_cast<uintptr_t>(v)) & 15)
        return FASTMAX_E_ALIGNMENT;
    if ((ldx * 2) & 15) return FASTMAX_E_ALIGNMENT;
    if (bias && (reinterpret_cast<uintptr_t>(bias) & 15)) return FASTMAX_E_ALIGNMENT;
    if (ea && ((reinterpret_cast<uintptr_t>(ea) | reinterpret_cast<uintptr_t>(eb)) & 15)) return FASTMAX_E_ALIGNMENT;
    GemmParams p{reinterpret_cast<const __bf16*>(x), w, GemmScale{nullptr, nullptr, nullptr, nullptr, 0.f}, bias,
                 reinterpret_cast<const __bf16*>(ea), reinterpret_cast<const __bf16*>(eb), nullptr, M, N, K, rank_pad, ldx, 0,
                 (N + 255) / 256, 0, 0, 0, cos, sin, q, k, v, T, G, q_per_kv, head_size, rope_n_elem, tables16 ? 1 : 0};
    return launch_gemm256a<7, true>(p, reinterpret_cast<hipStream_t>(stream));
}

// W^T [K][N] bf16 from the NF4 codes of W [N][K] (N % 64 == 0, K % 64 == 0, 16-byte aligned)
int fastmax_hip_nf4_dequantize_transposed(const uint8_t* wq, const fastmax_nf4_scales* scales, void* out, int N, int K, void* stream) {
    if (!wq || !scales || !out) return FASTMAX_E_NULL;
    if (scales->absmax_q ? !(scales->absmax2 && scales->code2) : !scales->absmax) return FASTMAX_E_NULL;
    if (N <= 0 || K <= 0 || (N % 64) || (K % 64)) return FASTMAX_E_BAD_SHAPE;
    if ((reinterpret_cast<uintptr_t>(wq) | reinterpret_cast<uintptr_t>(out)) & 15) return FASTMAX_E_ALIGNMENT;
    const GemmScale sc{scales->absmax, scales->absmax_q, scales->absmax2, scales->code2, scales->offset};
    hipLaunchKernelGGL(nf4_dequant_transposed_kernel, dim3(K / 64, N / 64), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), wq, sc,
                       reinterpret_cast<__bf16*>(out), N, K);
    return (int)hipGetLastError();
}

}  // extern "C"
