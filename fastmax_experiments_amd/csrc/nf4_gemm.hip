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
    // lm-head + cross entropy epilogues (EPI 1 / 2): the tile is a block of logits that never leaves the chip
    const int64_t* targets;   // [M]
    float* part;              // EPI 1: [M][nbn] (max, sum exp) of the row over this column block
    float* ztgt;              // EPI 1: [M] the target's logit (written by the one lane that holds it)
    const float* lse;         // EPI 2: [M] log sum exp of the row
    float gscale;             // EPI 2: d(loss)/d(row loss)
    int64_t ignore_index;
    // RoPE + QKV de-interleave epilogue (qlora_gemm256a_kernel<.., true>): the product is the qkv projection of a batch of
    // sequences (rows = b * T + t; columns = (group, slot, d) of lit_gpt/model.py:397-420) and leaves as q (B, G qpk, T, hs),
    // k, v (B, G, T, hs) with the rotation of model.py:702-708 applied to the first rope_n elements of the q and k heads
    const float *rope_cos, *rope_sin;   // (T, rope_n) float32
    void *rq, *rk, *rv;
    int T, G, qpk, hs, rope_n, tables16;
    unsigned long long* stamps;   // diagnostics: [workgroup][4] = main loop (shader cycles, 100 MHz ticks), wave 0's wait for its copies, for the barrier; null normally
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

// PF (dense weight only): every thread touches one 128-byte line of tile kt+2 (x rows / W rows) with a plain load that nobody
// waits for, so the LDS-DMA of that tile, issued a step later, is served from L2 instead of HBM; the loop then waits with a
// counted vmcnt (the prefetch stays in flight across the raw s_barrier) instead of the vmcnt(0) a __syncthreads() implies.
// ILV: the next tile's staging issued piece by piece BETWEEN the groups of matrix instructions (see mma_k32_with); measured
// neutral to 5 % slower than issuing it in front of the step ("gemm_sched" 6, kept for A/B)
// EPI: what happens to the finished 256 x 256 tile.  0: stored (the linear layer).  1 / 2: the tile is a block of lm-head
// logits z = x W^T, rounded to bf16 like the reference's head output, and consumed in registers --
//   1 (loss forward) : per row the (max, sum exp) pair over the block's columns -> part[m][bn]; the target's logit -> ztgt[m]
//   2 (loss backward): dz = (exp(z - lse_m) - [n == t_m]) * gscale (0 for rows that are not scored) stored as bf16
//   3 (loss forward that keeps the logits for the backward pass): 1, then the tile stored as bf16 like 0
// so the (tokens x vocabulary) logits are never written or read (lit_gpt/utils.py:228-272 after lora.py:547-550).
template <bool WNF4, bool HALVES, bool PF = false, bool ILV = false, int EPI = 0, bool REG = false, bool PIPE = false, bool SPEC = false, bool STAG = false>
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
    bool dma_all_waves = false;
    auto dma_tile = [&](const __bf16* base, int64_t ld, int row0, int nrows, int k0, char* dst) {
        if constexpr (SPEC) {
            // only one wave of each SIMD pair (w < 4; w and w + 4 share a SIMD) issues the copies, eight pieces per operand:
            // its partner goes straight to the matrix instructions, so the SIMD multiplies while the copies are being issued
            // (SPEC + STAG, experiment: the weight tile's pieces are shared by all eight waves, 12 / 4 pieces per pair)
            if (STAG && dma_all_waves) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int pc = 4 * w + j, row = 8 * pc + drow;
                    const int gr = min(row0 + row, nrows - 1);
                    const __bf16* src = base + (int64_t)gr * ld + k0 + 8 * dchunk;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)(dst + pc * 1024), 16, 0, 0);
                }
                return;
            }
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
            return;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = 8 * (4 * w + j) + drow;
            const int gr = min(row0 + row, nrows - 1);               // rows past the end re-read the last row (never stored)
            const __bf16* src = base + (int64_t)gr * ld + k0 + 8 * dchunk;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(dst + (4 * w + j) * 1024), 16, 0, 0);
        }
    };
    // REG: the same tiles through registers instead -- four 16-byte loads per operand per thread at the start of a step, four
    // ds_write_b128 into the swizzled image after the step's matrix instructions.  An LDS-DMA piece costs its wave 60-185
    // issue cycles (MI355X_MICROARCH.md), 16 of them per SIMD and step against 2048 cycles of matrix work; a plain load and
    // a ds_write_b128 cost ~4 + 13.
    gu32x4 xs[4], wsr[4];
    auto reg_load = [&](const __bf16* base, int64_t ld, int row0, int nrows, int k0, gu32x4 (&rg)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = 8 * (4 * w + j) + drow;
            const int gr = min(row0 + row, nrows - 1);
            rg[j] = *reinterpret_cast<const gu32x4*>(base + (int64_t)gr * ld + k0 + 8 * dslot);
        }
    };
    auto reg_store = [&](char* dst, const gu32x4 (&rg)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<gu32x4*>(dst + gsw(8 * (4 * w + j) + drow, dslot)) = rg[j];
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

    // one LDS-DMA instruction (1 KB = 8 rows) of a tile: the j-th of this wave's four
    auto dma_piece = [&](const __bf16* base, int64_t ld, int row0, int nrows, int k0, char* dst, int j) {
        const int row = 8 * (4 * w + j) + drow;
        const int gr = min(row0 + row, nrows - 1);
        const __bf16* src = base + (int64_t)gr * ld + k0 + 8 * dchunk;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(dst + (4 * w + j) * 1024), 16, 0, 0);
    };
    // a 32-deep step whose four groups of eight MFMAs are each followed by one piece of the NEXT tile's staging (an LDS-DMA
    // instruction or one decoded code word): issued between the matrix instructions their cost hides in the matrix pipe's
    // shadow; eight DMA issues in front of the step cost 500-1000 cycles of a 3500-cycle step (MI355X_MICROARCH.md: 60-185 each)
    auto mma_k32_with = [&](const char* Xs, const char* Ws, int ks, auto&& between) {
        gbf16x8 af[4], bfm[8];
#pragma unroll
        for (int t = 0; t < 4; ++t) af[t] = *reinterpret_cast<const gbf16x8*>(Ws + gsw(64 * wn + 16 * t + r, 4 * ks + q4));
#pragma unroll
        for (int t = 0; t < 8; ++t) bfm[t] = *reinterpret_cast<const gbf16x8*>(Xs + gsw(128 * wm + 16 * t + r, 4 * ks + q4));
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
            for (int mt = 0; mt < 8; ++mt)
                acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[nt], bfm[mt], acc[nt][mt], 0, 0, 0);
            between(nt);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
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
    auto mma_frags = [&](const gbf16x8 (&af)[4], const gbf16x8 (&bfm)[8]) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int mt = 0; mt < 8; ++mt)
                acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[nt], bfm[mt], acc[nt][mt], 0, 0, 0);
    };

    const int KT = K / BK;
    unsigned long long t0c = 0, t0r = 0, wait_dma = 0, wait_bar = 0;
    if (prm.stamps) { t0c = __builtin_amdgcn_s_memtime(); t0r = __builtin_amdgcn_s_memrealtime(); }
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
    // PF: the line this thread touches in every tile (threads 0..255: x row, 256..511: W row)
    unsigned int pf_sink = 0;
    const char* pf_base = nullptr;
    if constexpr (PF) {
        const int prow = tid & 255;
        pf_base = tid < 256 ? reinterpret_cast<const char*>(prm.x + (int64_t)min(m0 + prow, M - 1) * prm.ldx)
                            : reinterpret_cast<const char*>(reinterpret_cast<const __bf16*>(prm.w) + (int64_t)min(n0 + prow, N - 1) * K);
    }
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
                else {
                    dma_all_waves = true;
                    dma_tile(reinterpret_cast<const __bf16*>(prm.w), K, n0, N, (kt + 2) * BK, cur + XT);
                    dma_all_waves = false;
                }
            }
            const char* nf = more ? nxt : cur;                       // last step: a harmless re-read instead of a branch per tile
            half(a1, a0, nf, nf + XT, 0, true);
        }
        __syncthreads();
    } else if constexpr (STAG) {
        // the two waves of a SIMD (w, w + 4) leave the barrier together; with the same program both would issue copies, then
        // read fragments, then multiply -- and the matrix pipe idles while both are in their loading part (stamps: 1200 of
        // 3300 cycles per step).  Here waves 0-3 issue their copies of tile kt+1 first, waves 4-7 after their first 32-deep
        // half: one of the pair is always multiplying.
        const bool late = w >= 4;
        for (int kt = 0; kt < KT; ++kt) {
            char* cur = smem + (kt & 1) * STAGE;
            char* nxt = smem + ((kt & 1) ^ 1) * STAGE;
            const bool more = kt + 1 < KT;
            if (more && !late) {
                dma_tile(prm.x, prm.ldx, m0, M, (kt + 1) * BK, nxt);
                if constexpr (!WNF4) dma_tile(reinterpret_cast<const __bf16*>(prm.w), K, n0, N, (kt + 1) * BK, nxt + XT);
            }
            __builtin_amdgcn_sched_barrier(0);
            mma_k32(cur, cur + XT, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (more && late) {
                dma_tile(prm.x, prm.ldx, m0, M, (kt + 1) * BK, nxt);
                if constexpr (!WNF4) dma_tile(reinterpret_cast<const __bf16*>(prm.w), K, n0, N, (kt + 1) * BK, nxt + XT);
            }
            __builtin_amdgcn_sched_barrier(0);
            mma_k32(cur, cur + XT, 1);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (WNF4) {
                if (more) decode_codes(nxt + XT);
                if (kt + 2 < KT) load_codes((kt + 2) * BK);
            }
            if (prm.stamps) {
                const unsigned long long ta = __builtin_amdgcn_s_memtime();
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                const unsigned long long tb = __builtin_amdgcn_s_memtime();
                __builtin_amdgcn_s_barrier();
                wait_dma += tb - ta;
                wait_bar += __builtin_amdgcn_s_memtime() - tb;
            } else {
                __syncthreads();
            }
        }
    } else if constexpr (REG) {
        for (int kt = 0; kt < KT; ++kt) {
            char* cur = smem + (kt & 1) * STAGE;
            char* nxt = smem + ((kt & 1) ^ 1) * STAGE;
            const bool more = kt + 1 < KT;
            if (more) {
                reg_load(prm.x, prm.ldx, m0, M, (kt + 1) * BK, xs);
                if constexpr (!WNF4) reg_load(reinterpret_cast<const __bf16*>(prm.w), K, n0, N, (kt + 1) * BK, wsr);
            }
            __builtin_amdgcn_sched_barrier(0);
            mma_k32(cur, cur + XT, 0);
            mma_k32(cur, cur + XT, 1);
            __builtin_amdgcn_sched_barrier(0);
            if (more) {
                reg_store(nxt, xs);
                if constexpr (!WNF4) reg_store(nxt + XT, wsr);
            }
            if constexpr (WNF4) {
                if (more) decode_codes(nxt + XT);
                if (kt + 2 < KT) load_codes((kt + 2) * BK);
            }
            __syncthreads();
        }
    } else if constexpr (ILV) {
        for (int kt = 0; kt < KT; ++kt) {
            char* cur = smem + (kt & 1) * STAGE;
            char* nxt = smem + ((kt & 1) ^ 1) * STAGE;
            const bool more = kt + 1 < KT;
            const int k1 = (kt + 1) * BK;
            __builtin_amdgcn_sched_barrier(0);
            mma_k32_with(cur, cur + XT, 0, [&](int j) { if (more) dma_piece(prm.x, prm.ldx, m0, M, k1, nxt, j); });
            if constexpr (WNF4) {
                mma_k32_with(cur, cur + XT, 1, [&](int j) { if (more) decode_words(nxt + XT, j, j + 1); });
                if (kt + 2 < KT) load_codes((kt + 2) * BK);
            } else {
                mma_k32_with(cur, cur + XT, 1, [&](int j) {
                    if (more) dma_piece(reinterpret_cast<const __bf16*>(prm.w), K, n0, N, k1, nxt + XT, j);
                });
            }
            __syncthreads();
        }
    } else
    for (int kt = 0; kt < KT; ++kt) {
        char* cur = smem + (kt & 1) * STAGE;
        char* nxt = smem + ((kt & 1) ^ 1) * STAGE;
        const bool more = kt + 1 < KT;
        if (more) {
            dma_tile(prm.x, prm.ldx, m0, M, (kt + 1) * BK, nxt);
            if constexpr (!WNF4) dma_tile(reinterpret_cast<const __bf16*>(prm.w), K, n0, N, (kt + 1) * BK, nxt + XT);
        }
        if constexpr (PF) {
            // younger than the DMAs above on purpose: vmcnt(1) at the end of the step waits for them and not for this
            const char* pa = pf_base + (int64_t)min(kt + 2, KT - 1) * (BK * 2);
            asm volatile("global_load_dword %0, %1, off" : "+v"(pf_sink) : "v"(pa) : "memory");
        }
        if constexpr (WNF4) {
            // The decode of tile kt+1 (its codes were fetched a step ago) is vector + LDS work, the 64 MFMAs of tile kt are
            // matrix work: the two waves that share a SIMD (w and w + 4) run them in OPPOSITE order, so one decodes in the
            // shadow of the other's matrix instructions.  (HALVES = false: every wave decodes after its MFMAs, for A/B.)
            const bool decode_first = HALVES && w >= 4;
            if (decode_first) {                                      // (the matrix code below is common to both halves: no
                if (more) decode_codes(nxt + XT);                    //  accumulator merges across a branch)
                if (kt + 2 < KT) load_codes((kt + 2) * BK);
            }
            __builtin_amdgcn_sched_barrier(0);
            mma_k32(cur, cur + XT, 0);
            mma_k32(cur, cur + XT, 1);
            __builtin_amdgcn_sched_barrier(0);
            if (!decode_first) {
                if (more) decode_codes(nxt + XT);
                if (kt + 2 < KT) load_codes((kt + 2) * BK);
            }
        } else {
            mma_k32(cur, cur + XT, 0);
            mma_k32(cur, cur + XT, 1);
        }
        if constexpr (PF) {
            asm volatile("s_waitcnt vmcnt(1)" ::: "memory");         // this wave's DMAs of tile kt+1 landed; the prefetch may still fly
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        } else if (prm.stamps) {
            // diagnostics: how long this wave waits for its own copies of tile kt+1, and then for the other waves
            const unsigned long long ta = __builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            const unsigned long long tb = __builtin_amdgcn_s_memtime();
            __builtin_amdgcn_s_barrier();
            const unsigned long long tc = __builtin_amdgcn_s_memtime();
            wait_dma += tb - ta;
            wait_bar += tc - tb;
        } else {
            __syncthreads();                                         // tile kt+1 landed (DMA drained by the barrier's wait), tile kt consumed
        }
    }
    if constexpr (PF) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("" ::"v"(pf_sink));
    }
    if (prm.stamps && tid == 0) {
        prm.stamps[4 * blockIdx.x] = __builtin_amdgcn_s_memtime() - t0c;
        prm.stamps[4 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - t0r;
        prm.stamps[4 * blockIdx.x + 2] = wait_dma;
        prm.stamps[4 * blockIdx.x + 3] = wait_bar;
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
    if constexpr (EPI == 1 || EPI == 3) {
        // ---- loss forward: nothing of the tile is stored (EPI 3: the reduction first, then the tile is stored as well).  Lane (r, q4) of wave (wm, wn) holds, for row 128 wm + 16 mt + r,
        //      the 16 columns 64 wn + 16 nt + 4 q4 + i: reduce in the lane, across the four q4 lanes, then across the four wn waves
        float2* red = reinterpret_cast<float2*>(smem);               // [256 m][4 wn]
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
            const int m = 128 * wm + 16 * mt + r, gm = m0 + m;
            const int64_t t = gm < M ? prm.targets[gm] : -1;
            const int64_t tloc = t - n0 - 64 * wn;                    // the target's column inside this wave's 64, if any
            float z[16], mx = -INFINITY;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int n = 16 * nt + 4 * q4 + i;
                    float v = (float)(__bf16)acc[nt][mt][i];
                    if (n0 + 64 * wn + n >= N) v = -INFINITY;
                    else if (n == tloc && t != prm.ignore_index) prm.ztgt[gm] = v;
                    z[4 * nt + i] = v;
                    mx = fmaxf(mx, v);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            float sum = 0.f;
            if (mx > -INFINITY) {
#pragma unroll
                for (int e = 0; e < 16; ++e) sum += __expf(z[e] - mx);
            }
            sum += __shfl_xor(sum, 16, 64);
            sum += __shfl_xor(sum, 32, 64);
            if (q4 == 0) red[m * 4 + wn] = make_float2(mx, sum);
        }
        __syncthreads();
        if (tid < BM && m0 + tid < M) {
            float mx = -INFINITY, sum = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float2 pr = red[tid * 4 + j];
                const float mm = fmaxf(mx, pr.x);
                const float a = sum > 0.f ? sum * __expf(mx - mm) : 0.f, b = pr.y > 0.f ? pr.y * __expf(pr.x - mm) : 0.f;
                mx = mm;
                sum = a + b;
            }
            reinterpret_cast<float2*>(prm.part)[(int64_t)(m0 + tid) * prm.nbn + bn] = make_float2(mx, sum);
        }
        if constexpr (EPI == 1) return;
        __syncthreads();                                             // `red` is about to be overwritten by the tile image
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
        float row_lse = 0.f, row_scale = 0.f;
        int64_t tloc = -1;
        if constexpr (EPI == 2) {
            const int gm = min(m0 + m, M - 1);
            const int64_t t = prm.targets[gm];
            row_lse = prm.lse[gm];
            row_scale = (t != prm.ignore_index && t >= 0 && t < N) ? prm.gscale : 0.f;
            tloc = t - n0;
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int n = 64 * wn + 16 * nt + 4 * q4;                // 4 consecutive columns = 8 bytes
            gf32x4 v = acc[nt][mt] + bias4[nt];
            if constexpr (EPI == 2) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float p = __expf((float)(__bf16)v[i] - row_lse) - ((n + i == tloc) ? 1.f : 0.f);
                    v[i] = p * row_scale;
                }
            }
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

// ---- experiment ("gemm_sched" 20): the same 256 x 256 x 64 tile with FOUR waves, each owning 128 x 128 (256 accumulator
// registers: the unified 512-register file at one wave per SIMD), operands staged through registers (global_load_dwordx4 ->
// ds_write_b128) and every load / store / fragment read interleaved with the matrix instructions of the same wave:
//   half 0 of step kt: 64 MFMAs on fragments F0 | the 16 ds_writes of tile kt+1 (loaded during the previous half) + the 16
//                      fragment reads F1 of this tile's second 32-deep half
//   barrier            tile kt+1 whole in the other stage
//   half 1:            64 MFMAs on F1 | the 16 global loads of tile kt+2 + the 16 fragment reads F0 of tile kt+1
// Fewer LDS bytes per flop than the 8-wave form (16 fragment reads per 64 MFMAs instead of 12 per 32) and no LDS-DMA issue cost;
// dense bf16 weight only.
namespace g256w4 {
constexpr int BM = 256, BN = 256, BK = 64;
constexpr int XT = BM * BK * 2, STAGE = 2 * XT;
constexpr int LDS_BYTES = 2 * STAGE;                                         // 131072
}  // namespace g256w4

__global__ __launch_bounds__(256, 1) void qlora_gemm256w4_kernel(GemmParams prm) {
    using namespace g256w4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 1, wn = w & 1;
    const int r = lane & 15, q4 = lane >> 4;
    int bn, bm;
    gemm_tile_of(prm, bm, bn);
    const int m0 = bm * BM, n0 = bn * BN;
    const int M = prm.M, N = prm.N, K = prm.K;
    // staging map: piece j of a thread = 16-byte chunk (tid & 7) of tile row (tid >> 3) + 32 j.  Whole tiles only (the host sends
    // ragged M or N to the 8-wave kernel), so a piece's address is a workgroup-uniform base + ONE per-thread offset: no
    // per-piece address registers (the 512-register budget is 256 accumulators + 96 fragment + 64 staging registers).
    const int srow = tid >> 3, sch = tid & 7;
    const unsigned xoff = (unsigned)(srow * (int)prm.ldx + 8 * sch) * 2u, woff = (unsigned)(srow * K + 8 * sch) * 2u;
    const char* xblk = reinterpret_cast<const char*>(prm.x + (int64_t)m0 * prm.ldx);
    const char* wblk = reinterpret_cast<const char*>(reinterpret_cast<const __bf16*>(prm.w) + (int64_t)n0 * K);
    const int64_t xstep = 64 * prm.ldx, wstep = 64 * (int64_t)K;     // bytes between pieces (32 rows)
    const int sdst0 = gsw(srow, sch);                                // piece j lands 32 rows = 4096 bytes further (same swizzle)
    gu32x4 rx[8], rw[8];
    auto load_piece = [&](int k0, int j) {                           // j: 0..7 x, 8..15 W
        if (j < 8) rx[j] = *reinterpret_cast<const gu32x4*>(xblk + j * xstep + 2 * k0 + xoff);
        else rw[j - 8] = *reinterpret_cast<const gu32x4*>(wblk + (j - 8) * wstep + 2 * k0 + woff);
    };
    auto store_piece = [&](char* stage, int j) {
        if (j < 8) *reinterpret_cast<gu32x4*>(stage + sdst0 + 4096 * j) = rx[j];
        else *reinterpret_cast<gu32x4*>(stage + XT + sdst0 + 4096 * (j - 8)) = rw[j - 8];
    };
    gf32x4 acc[8][8];                                                // [nt][mt]
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = gf32x4{0, 0, 0, 0};
    gbf16x8 a0[8], a1[8], b[8];                                      // weight fragments double-buffered, x fragments rotate in place
    auto frag_a = [&](const char* st, int ks, int t) {
        return *reinterpret_cast<const gbf16x8*>(st + XT + gsw(128 * wn + 16 * t + r, 4 * ks + q4));
    };
    auto frag_b = [&](const char* st, int ks, int t) {
        return *reinterpret_cast<const gbf16x8*>(st + gsw(128 * wm + 16 * t + r, 4 * ks + q4));
    };
    // one 32-deep half: row tile by row tile (mt), 8 MFMAs each.  After row tile mt: b[mt] is refilled with the next half's
    // fragment and ONE weight fragment of the next half is fetched into the other buffer (so the first row tile after the
    // barrier finds all its operands long since loaded); `side(2 mt)`, `side(2 mt + 1)` carry the staging traffic.
    // The MFMAs are in place in the accumulation registers by inline asm: left to the register allocator, the two unrolled
    // halves get differently numbered accumulators and ~300 v_accvgpr_mov / read / write per K step to line them up again.
    // With one wave per SIMD nothing else fills the gaps: a 16-cycle MFMA leaves 8 issue cycles, so the side instructions are
    // spread ONE per pair of MFMAs (clustered behind a row tile they delayed the next MFMA by their whole issue time:
    // stamps gave 22-27 cycles per MFMA).  b[mt] can only be refilled once its row tile is done: it is refilled two MFMAs into
    // the NEXT row tile.
    auto half = [&](const gbf16x8 (&af)[8], gbf16x8 (&afn)[8], const char* nst, int nks, auto&& side) {
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
#pragma unroll
            for (int nt = 0; nt < 8; ++nt) {
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[nt][mt]) : "v"(af[nt]), "v"(b[mt]));
                if (nt == 1) {
                    if (mt > 0) b[mt - 1] = frag_b(nst, nks, mt - 1);
                    __builtin_amdgcn_sched_barrier(0);
                } else if (nt == 3) {
                    afn[mt] = frag_a(nst, nks, mt);
                    __builtin_amdgcn_sched_barrier(0);
                } else if (nt == 5) {
                    side(2 * mt);
                    __builtin_amdgcn_sched_barrier(0);
                } else if (nt == 7) {
                    side(2 * mt + 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        b[7] = frag_b(nst, nks, 7);
    };
    const int KT = K / BK;
    unsigned long long t0c = 0, t0r = 0, tl = 0, st_h0 = 0, st_h1 = 0, st_wait = 0, st_bar = 0;
    if (prm.stamps) { t0c = __builtin_amdgcn_s_memtime(); t0r = __builtin_amdgcn_s_memrealtime(); }
    // prologue: tile 0 through registers into stage 0, tile 1 into registers
#pragma unroll
    for (int j = 0; j < 16; ++j) load_piece(0, j);
#pragma unroll
    for (int j = 0; j < 16; ++j) store_piece(smem, j);
#pragma unroll
    for (int j = 0; j < 16; ++j) load_piece(min(1, KT - 1) * BK, j);
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 8; ++t) { a0[t] = frag_a(smem, 0, t); b[t] = frag_b(smem, 0, t); }
    if (prm.stamps) tl = __builtin_amdgcn_s_memtime();
    for (int kt = 0; kt < KT; ++kt) {
        char* cur = smem + (kt & 1) * STAGE;
        char* nxt = smem + ((kt & 1) ^ 1) * STAGE;
        const bool more = kt + 1 < KT;
        // half 0: F0 = a0; tile kt+1 (in registers) -> nxt; F1 <- cur, second half.  No branches inside the halves: on the last
        // steps the stores go to a stage nobody reads again and the loads re-read the last tile.
        half(a0, a1, cur, 1, [&](int j) { store_piece(nxt, j); });
        if (prm.stamps) {
            const unsigned long long ta = __builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            const unsigned long long tb = __builtin_amdgcn_s_memtime();
            __builtin_amdgcn_s_barrier();
            const unsigned long long tc = __builtin_amdgcn_s_memtime();
            st_h0 += ta - tl;  st_wait += tb - ta;  st_bar += tc - tb;  tl = tc;
        } else {
            __syncthreads();
        }
        // half 1: F1 = a1; tile kt+2 -> registers; F0 <- nxt, first half (a harmless re-read of cur on the last step)
        const char* nf = more ? nxt : cur;
        const int k2 = min(kt + 2, KT - 1) * BK;
        half(a1, a0, nf, 0, [&](int j) { load_piece(k2, j); });
        if (prm.stamps) { const unsigned long long td = __builtin_amdgcn_s_memtime(); st_h1 += td - tl; tl = td; }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");                // the hazard recogniser does not see inside the asm MFMAs
    if (prm.stamps && tid == 0) {
        prm.stamps[4 * blockIdx.x] = __builtin_amdgcn_s_memtime() - t0c;
        prm.stamps[4 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - t0r;
        prm.stamps[4 * blockIdx.x + 2] = st_wait + (st_h0 << 32);   // low word: wait before the barrier; high word: half 0
        prm.stamps[4 * blockIdx.x + 3] = st_bar + (st_h1 << 32);     // low word: barrier; high word: half 1
    }
    __syncthreads();
    // ---- LoRA branch: one more 32-deep step over the padded rank --------------------------------------------------------------
    if (prm.ea && prm.eb) {
        const int cpr = prm.RP / 8;
        const gbf16x8 z = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
        for (int i = tid; i < BM * 4; i += 256) {
            const int row = i >> 2, c = i & 3;
            const int gm = min(m0 + row, M - 1), gn = min(n0 + row, N - 1);
            *reinterpret_cast<gbf16x8*>(smem + gsw(row, c)) = c < cpr ? *reinterpret_cast<const gbf16x8*>(prm.ea + (int64_t)gm * prm.RP + 8 * c) : z;
            *reinterpret_cast<gbf16x8*>(smem + XT + gsw(row, c)) = c < cpr ? *reinterpret_cast<const gbf16x8*>(prm.eb + (int64_t)gn * prm.RP + 8 * c) : z;
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 8; ++t) { a0[t] = frag_a(smem, 0, t); b[t] = frag_b(smem, 0, t); }
#pragma unroll
        for (int nt = 0; nt < 8; ++nt)
#pragma unroll
            for (int mt = 0; mt < 8; ++mt)
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[nt][mt]) : "v"(a0[nt]), "v"(b[mt]));
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        __syncthreads();
    }
    // ---- epilogue: tile -> LDS as [256 m][256 n] bf16, then whole rows out -------------------------------------------------------
    char* ct = smem;
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) {
        const int gn = min(n0 + 128 * wn + 16 * nt + 4 * q4, N - 4);
        const gf32x4 bias4 = prm.bias ? *reinterpret_cast<const gf32x4*>(prm.bias + gn) : gf32x4{0, 0, 0, 0};
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
            const int m = 128 * wm + 16 * mt + r, n = 128 * wn + 16 * nt + 4 * q4;
            const gf32x4 v = acc[nt][mt] + bias4;
            gbf16x4 o;
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = (__bf16)v[i];
            const int chunk = n >> 3;
            *reinterpret_cast<gbf16x4*>(ct + m * 512 + (((chunk ^ m) & 31) << 4) + ((n & 4) << 1)) = o;
        }
    }
    __syncthreads();
    for (int i = tid; i < BM * 32; i += 256) {
        const int m = i >> 5, chunk = i & 31;
        const int gm = m0 + m, gn = n0 + 8 * chunk;
        if (gm < M && gn < N) {
            const gu32x4 v = *reinterpret_cast<const gu32x4*>(ct + m * 512 + (((chunk ^ m) & 31) << 4));
            *reinterpret_cast<gu32x4*>(prm.y + (int64_t)gm * prm.ldy + gn) = v;
        }
    }
}

// ---- experiment ("gemm_sched" 21): the four-wave form on 32 x 32 x 16 MFMAs.  A 32-cycle matrix instruction leaves 24 issue
// cycles for one ds_write_b128 (13) or a fragment read / global load, where the 16-cycle one leaves 8 and every longer side
// instruction pushes the next MFMA back.  Images: 128 data bytes + 16 pad bytes per row (conflict-free for 32-row ds_read_b128
// fragments, which the XOR image is not).  A K step = four 16-deep sub-steps s0..s3 of 16 MFMAs; fragments of sub-step s+1 are
// read during s; the 16 stores of tile kt+1 are spread over s0..s2, the barrier sits between s2 and s3 (all reads of tile kt are
// issued before it, so its stage is free for tile kt+2 afterwards), s3 reads tile kt+1's first fragments and issues the 16 loads
// of tile kt+2.
typedef float gf32x16 __attribute__((ext_vector_type(16)));
namespace g256w4b {
constexpr int BM = 256, BN = 256, BK = 64, ROWB = 144;
constexpr int XT = BM * ROWB, STAGE = 2 * XT;                                // 36864, 73728
constexpr int LDS_BYTES = 2 * STAGE;                                         // 147456 (the epilogue image needs 131072)
}  // namespace g256w4b

__global__ __launch_bounds__(256, 1) void qlora_gemm256w4b_kernel(GemmParams prm) {
    using namespace g256w4b;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 1, wn = w & 1;
    const int l31 = lane & 31, h = lane >> 5;
    int bn, bm;
    gemm_tile_of(prm, bm, bn);
    const int m0 = bm * BM, n0 = bn * BN;
    const int N = prm.N, K = prm.K;
    const int srow = tid >> 3, sch = tid & 7;
    const unsigned xoff = (unsigned)(srow * (int)prm.ldx + 8 * sch) * 2u, woff = (unsigned)(srow * K + 8 * sch) * 2u;
    const char* xblk = reinterpret_cast<const char*>(prm.x + (int64_t)m0 * prm.ldx);
    const char* wblk = reinterpret_cast<const char*>(reinterpret_cast<const __bf16*>(prm.w) + (int64_t)n0 * K);
    const int64_t xstep = 64 * prm.ldx, wstep = 64 * (int64_t)K;
    const int sdst0 = srow * ROWB + 16 * sch;                        // piece j: 32 rows = 32 * 144 bytes further
    gu32x4 rx[8], rw[8];
    auto load_piece = [&](int k0, int j) {
        if (j < 8) rx[j] = *reinterpret_cast<const gu32x4*>(xblk + j * xstep + 2 * k0 + xoff);
        else rw[j - 8] = *reinterpret_cast<const gu32x4*>(wblk + (j - 8) * wstep + 2 * k0 + woff);
    };
    auto store_piece = [&](char* stage, int j) {
        if (j < 8) *reinterpret_cast<gu32x4*>(stage + sdst0 + 32 * ROWB * j) = rx[j];
        else *reinterpret_cast<gu32x4*>(stage + XT + sdst0 + 32 * ROWB * (j - 8)) = rw[j - 8];
    };
    gf32x16 acc[4][4];                                               // [nt][mt]: C rows = n (registers), column = m (lane & 31)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    // fragment of 16-deep sub-step s (k = 16 s + 8 h .. + 7): row 32 t + l31 of the wave's 128 rows
    const int fa0 = XT + (128 * wn + l31) * ROWB + 16 * h, fb0 = (128 * wm + l31) * ROWB + 16 * h;
    auto frag_a = [&](const char* st, int s_, int t) { return *reinterpret_cast<const gbf16x8*>(st + fa0 + 32 * ROWB * t + 32 * s_); };
    auto frag_b = [&](const char* st, int s_, int t) { return *reinterpret_cast<const gbf16x8*>(st + fb0 + 32 * ROWB * t + 32 * s_); };
    gbf16x8 fa[2][4], fb[2][4];
    // one sub-step: 16 MFMAs on fragment set `c`; gap g (after MFMA g) carries `gap(g)`
    auto substep = [&](int c, auto&& gap) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[nt][mt]) : "v"(fa[c][nt]), "v"(fb[c][mt]));
                gap(4 * mt + nt);
                __builtin_amdgcn_sched_barrier(0);
            }
    };
    const int KT = K / BK;
    unsigned long long t0c = 0, t0r = 0;
    if (prm.stamps) { t0c = __builtin_amdgcn_s_memtime(); t0r = __builtin_amdgcn_s_memrealtime(); }
#pragma unroll
    for (int j = 0; j < 16; ++j) load_piece(0, j);
#pragma unroll
    for (int j = 0; j < 16; ++j) store_piece(smem, j);
#pragma unroll
    for (int j = 0; j < 16; ++j) load_piece(min(1, KT - 1) * BK, j);
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 4; ++t) { fa[0][t] = frag_a(smem, 0, t); fb[0][t] = frag_b(smem, 0, t); }
    for (int kt = 0; kt < KT; ++kt) {
        char* cur = smem + (kt & 1) * STAGE;
        char* nxt = smem + ((kt & 1) ^ 1) * STAGE;
        const char* nf = (kt + 1 < KT) ? nxt : cur;
        const int k2 = min(kt + 2, KT - 1) * BK;
        // s0: set 0; reads set 1 <- (cur, 1) in gaps 0..7 (even ones: a, odd: b); stores 0..5 in gaps 8..13
        substep(0, [&](int g) {
            if (g < 8) { if (g & 1) fb[1][g >> 1] = frag_b(cur, 1, g >> 1); else fa[1][g >> 1] = frag_a(cur, 1, g >> 1); }
            else if (g < 14) store_piece(nxt, g - 8);
        });
        // s1: set 1; reads set 0 <- (cur, 2); stores 6..10
        substep(1, [&](int g) {
            if (g < 8) { if (g & 1) fb[0][g >> 1] = frag_b(cur, 2, g >> 1); else fa[0][g >> 1] = frag_a(cur, 2, g >> 1); }
            else if (g < 13) store_piece(nxt, g - 2);
        });
        // s2: set 0; reads set 1 <- (cur, 3); stores 11..15
        substep(0, [&](int g) {
            if (g < 8) { if (g & 1) fb[1][g >> 1] = frag_b(cur, 3, g >> 1); else fa[1][g >> 1] = frag_a(cur, 3, g >> 1); }
            else if (g < 13) store_piece(nxt, g + 3);
        });
        __syncthreads();                                             // tile kt+1 whole; every read of tile kt has been issued
        // s3: set 1; reads set 0 <- (tile kt+1, 0); the 16 loads of tile kt+2, one per gap
        substep(1, [&](int g) {
            if (g < 8) { if (g & 1) fb[0][g >> 1] = frag_b(nf, 0, g >> 1); else fa[0][g >> 1] = frag_a(nf, 0, g >> 1); }
            load_piece(k2, g);
        });
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    if (prm.stamps && tid == 0) {
        prm.stamps[4 * blockIdx.x] = __builtin_amdgcn_s_memtime() - t0c;
        prm.stamps[4 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - t0r;
        prm.stamps[4 * blockIdx.x + 2] = 0;
        prm.stamps[4 * blockIdx.x + 3] = 0;
    }
    __syncthreads();
    // ---- LoRA branch: rank padded to 16 or 32 = one or two 16-deep sub-steps ------------------------------------------------------
    if (prm.ea && prm.eb) {
        const int cpr = prm.RP / 8;
        for (int i = tid; i < BM * cpr; i += 256) {
            const int row = i / cpr, c = i % cpr;
            *reinterpret_cast<gbf16x8*>(smem + row * ROWB + 16 * c) = *reinterpret_cast<const gbf16x8*>(prm.ea + (int64_t)(m0 + row) * prm.RP + 8 * c);
            *reinterpret_cast<gbf16x8*>(smem + XT + row * ROWB + 16 * c) = *reinterpret_cast<const gbf16x8*>(prm.eb + (int64_t)(n0 + row) * prm.RP + 8 * c);
        }
        __syncthreads();
        for (int s_ = 0; s_ < prm.RP / 16; ++s_) {
#pragma unroll
            for (int t = 0; t < 4; ++t) { fa[0][t] = frag_a(smem, s_, t); fb[0][t] = frag_b(smem, s_, t); }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[nt][mt]) : "v"(fa[0][nt]), "v"(fb[0][mt]));
        }
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        __syncthreads();
    }
    // ---- epilogue: tile -> LDS as [256 m][256 n] bf16 (512-byte rows, chunk index XOR-ed with m & 31), then whole rows out --------
    char* ct = smem;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {                             // accumulator elements 4 gq .. + 3: rows n = 8 gq + 4 h + 0..3
            const int n = 128 * wn + 32 * nt + 8 * gq + 4 * h;
            const gf32x4 bias4 = prm.bias ? *reinterpret_cast<const gf32x4*>(prm.bias + n0 + n) : gf32x4{0, 0, 0, 0};
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int m = 128 * wm + 32 * mt + l31;
                gbf16x4 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = (__bf16)(acc[nt][mt][4 * gq + i] + bias4[i]);
                const int chunk = n >> 3;
                *reinterpret_cast<gbf16x4*>(ct + m * 512 + (((chunk ^ m) & 31) << 4) + ((n & 4) << 1)) = o;
            }
        }
    __syncthreads();
    for (int i = tid; i < BM * 32; i += 256) {
        const int m = i >> 5, chunk = i & 31;
        const gu32x4 v = *reinterpret_cast<const gu32x4*>(ct + m * 512 + (((chunk ^ m) & 31) << 4));
        *reinterpret_cast<gu32x4*>(prm.y + (int64_t)(m0 + m) * prm.ldy + n0 + 8 * chunk) = v;
    }
}

static int launch_gemm256w4b(GemmParams p, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(qlora_gemm256w4b_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           g256w4b::LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    p.nbm = (p.M + 255) / 256;
    gemm_map(p);
    hipLaunchKernelGGL(qlora_gemm256w4b_kernel, dim3(p.nbn * p.nbm), dim3(256), g256w4b::LDS_BYTES, stream, p);
    return (int)hipGetLastError();
}

static int launch_gemm256w4(GemmParams p, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(qlora_gemm256w4_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           g256w4::LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    p.nbm = (p.M + 255) / 256;
    gemm_map(p);
    hipLaunchKernelGGL(qlora_gemm256w4_kernel, dim3(p.nbn * p.nbm), dim3(256), g256w4::LDS_BYTES, stream, p);
    return (int)hipGetLastError();
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

// rows of the fused head loss: the per-column-block (max, sum exp) pairs of a row -> its log-sum-exp and loss
__global__ __launch_bounds__(256) void lmhead_ce_combine_kernel(const float* part, const float* ztgt, const int64_t* targets, float* loss,
                                                                float* lse, int M, int V, int nbn, int64_t ignore_index) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    const float2* pr = reinterpret_cast<const float2*>(part) + (int64_t)m * nbn;
    float mx = -INFINITY;
    for (int b = 0; b < nbn; ++b) mx = fmaxf(mx, pr[b].x);
    float sum = 0.f;
    for (int b = 0; b < nbn; ++b) sum += pr[b].y > 0.f ? pr[b].y * __expf(pr[b].x - mx) : 0.f;
    const float l = mx + __logf(sum);
    const int64_t t = targets[m];
    lse[m] = l;
    loss[m] = (t != ignore_index && t >= 0 && t < V) ? l - ztgt[m] : 0.f;
}

template <bool WNF4, bool HALVES, bool PF = false, bool ILV = false, int EPI = 0, bool REG = false, bool PIPE = false, bool SPEC = false, bool STAG = false>
static int launch_gemm256(GemmParams p, hipStream_t stream) {
    auto kern = qlora_gemm256_kernel<WNF4, HALVES, PF, ILV, EPI, REG, PIPE, SPEC, STAG>;
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

static unsigned long long* g_gemm_stamps = nullptr;

extern "C" {

// diagnostics (tools/gemm_clock.py): a device buffer of 4 x (number of workgroups) 64-bit words that the next
// fastmax_hip_qlora_gemm launches fill with the main loop's shader cycles and 100 MHz ticks per workgroup; null turns it off
void fastmax_hip_debug_gemm_stamps(void* buffer) { g_gemm_stamps = reinterpret_cast<unsigned long long*>(buffer); }

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
                 nullptr, nullptr, nullptr, nullptr, 0.f, 0, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0, 0, 0, g_gemm_stamps};
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    // dense weight: "gemm_sched" 5 = the L2-prefetch form (measured 5-9 % slower than the plain two-stage loop: kept for A/B)
    const int sched = tune_get(TUNE_GEMM_SCHED);
    // 256 x 256 tiles for at most half of the 256 CUs: 128-row tiles fill the chip in one round (measured: (4096, 2048, 2048)
    // 0.054 -> 0.044 ms, (2048, 2560, 2048) 0.052 -> 0.035; with 160 tiles, e.g. (4096, 2560, 2048), the 256-row tiles stay
    // ahead, 0.055 vs 0.068: two rounds of half tiles cost more than one round of whole ones).  Dense weight only;
    // "gemm_sched" 12 forces them, 13 forbids them.
    if (!w_is_nf4 && sched == 20 && (M % 256) == 0 && (N % 256) == 0) return launch_gemm256w4(p, st);
    if (!w_is_nf4 && sched == 21 && (M % 256) == 0 && (N % 256) == 0) return launch_gemm256w4b(p, st);
    if (!w_is_nf4 && sched != 13 && (sched == 12 || (int64_t)((M + 255) / 256) * ((N + 255) / 256) <= 128)) return launch_gemm128(p, st);
    if (sched == 16 && !w_is_nf4) return launch_gemm256a<7>(p, st);
    if (sched == 17 && !w_is_nf4) return launch_gemm256a<6>(p, st);
    if (sched == 18 && !w_is_nf4) return launch_gemm256a<5>(p, st);
    if (sched == 15 && !w_is_nf4) return launch_gemm256<false, false, false, false, 0, false, true, true, true>(p, st);
    if (sched == 11) return w_is_nf4 ? launch_gemm256<true, false, false, false, 0, false, false, false, true>(p, st)
                                     : launch_gemm256<false, false, false, false, 0, false, false, false, true>(p, st);
    if (sched == 9) return w_is_nf4 ? launch_gemm256<true, false, false, false, 0, false, false, true>(p, st)
                                    : launch_gemm256<false, false, false, false, 0, false, false, true>(p, st);
    if (sched == 10) return w_is_nf4 ? launch_gemm256<true, false, false, false, 0, false, true, true>(p, st)
                                     : launch_gemm256<false, false, false, false, 0, false, true, true>(p, st);
    if (sched == 8) return w_is_nf4 ? launch_gemm256<true, false, false, false, 0, false, true>(p, st)
                                    : launch_gemm256<false, false, false, false, 0, false, true>(p, st);
    if (sched == 7) return w_is_nf4 ? launch_gemm256<true, false, false, false, 0, true>(p, st)
                                    : launch_gemm256<false, false, false, false, 0, true>(p, st);
    if (!w_is_nf4) {
        if (sched == 5) return launch_gemm256<false, false, true>(p, st);
        if (sched == 6) return launch_gemm256<false, false, false, true>(p, st);
        if (sched == 14) return launch_gemm256<false, false, false>(p, st);       // round 2's first loop: every wave issues its copies, then reads, then multiplies
        // default (= "gemm_sched" 16): copies issued by one wave of each SIMD pair, fragments of the next half read under the
        // MFMAs of this one (5-9 % ahead of the plain loop at every fine-tune shape), and the copy-issuing wave of a pair
        // owning 7 of the strip's 16 row tiles, its partner 9 (another 2-3 %) -- profiles/r02_qlora_gemm.md; same bits
        return launch_gemm256a<7>(p, st);
    }
    if (sched == 6) return launch_gemm256<true, false, false, true>(p, st);
    // NF4 in the loop: "gemm_sched" 1 = SIMD partner waves decode / multiply in opposite order (measured 5-10 % slower), else
    // every wave decodes after its matrix instructions
    if (sched == 1) return launch_gemm256<true, true>(p, st);
    if (sched == 14) return launch_gemm256<true, false>(p, st);
    // fragments of the next half read under the MFMAs; every wave issues its own x copies (with the decode in the loop the
    // one-wave-per-pair copies measured no better: 0.217 vs 0.208 ms at (16384, 2560, 2048); plain loop 0.221)
    return launch_gemm256<true, false, false, false, 0, false, true>(p, st);
}

// ---- lm-head + cross entropy without the logits (finetune/lora.py:216-219 = GPT.forward's chunked head,
//      lora.py:547-550, feeding chunked_cross_entropy, lit_gpt/utils.py:228-272).  x [M][K] bf16 hidden states, w [V][K] bf16
//      head weight, targets [M] int64.  K % 64 == 0, V % 8 == 0, 16-byte aligned rows.
int64_t fastmax_hip_lmhead_ce_workspace(int M, int V) {
    if (M <= 0 || V <= 0) return 0;
    return ((int64_t)M * ((V + 255) / 256) * 2 + M) * (int64_t)sizeof(float);
}

// loss[m] = logsumexp_v(z_mv) - z_m,t(m) with z = bf16(x W^T) (0 for rows that are not scored: ignore_index or outside [0, V));
// lse[m] kept for the backward pass.  workspace: fastmax_hip_lmhead_ce_workspace(M, V) bytes.  `logits` non-null: the bf16
// logits are stored as well ([M][V], leading dimension ldz) -- the backward pass then needs no second product (288 GB of HBM
// hold them easily: 1 GB per 16384 x 32000), fastmax_hip_cross_entropy_backward in place + dx = dz . W.
int fastmax_hip_lmhead_ce_forward(const void* x, int64_t ldx, const void* w, const int64_t* targets, float* loss, float* lse,
                                  void* workspace, void* logits, int64_t ldz, int M, int V, int K, int64_t ignore_index,
                                  void* stream) {
    if (!x || !w || !targets || !loss || !lse || !workspace) return FASTMAX_E_NULL;
    if (M <= 0 || V <= 0 || K <= 0 || (K % 64) || (V % 8)) return FASTMAX_E_BAD_SHAPE;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(workspace)) & 15) return FASTMAX_E_ALIGNMENT;
    if ((ldx * 2) & 15) return FASTMAX_E_ALIGNMENT;
    if (logits && ((reinterpret_cast<uintptr_t>(logits) & 15) || ((ldz * 2) & 15))) return FASTMAX_E_ALIGNMENT;
    const int nbn = (V + 255) / 256;
    float* part = reinterpret_cast<float*>(workspace);
    float* ztgt = part + (int64_t)M * nbn * 2;
    GemmParams p{reinterpret_cast<const __bf16*>(x), w, GemmScale{nullptr, nullptr, nullptr, nullptr, 0.f}, nullptr, nullptr, nullptr,
                 reinterpret_cast<__bf16*>(logits), M, V, K, 0, ldx, ldz, nbn, 0, 0, 0, targets, part, ztgt, nullptr, 0.f, ignore_index,
                 nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0, 0, 0, nullptr};
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int rc = logits ? launch_gemm256<false, false, false, false, 3, false, true, true>(p, st)
                          : launch_gemm256<false, false, false, false, 1, false, true, true>(p, st);
    if (rc) return rc;
    hipLaunchKernelGGL(lmhead_ce_combine_kernel, dim3((M + 255) / 256), dim3(256), 0, st, part, ztgt, targets, loss, lse, M, V, nbn,
                       ignore_index);
    return (int)hipGetLastError();
}

// dz[m][v] = (exp(z_mv - lse[m]) - [v == t(m)]) * grad_scale as bf16 (rows that are not scored: 0), the logits recomputed in
// the tile and never stored: dx = dz . W and (a trainable head's) dW = dz^T . x are plain matrix products of the result.
int fastmax_hip_lmhead_ce_backward(const void* x, int64_t ldx, const void* w, const int64_t* targets, const float* lse,
                                   float grad_scale, void* dz, int64_t ldz, int M, int V, int K, int64_t ignore_index, void* stream) {
    if (!x || !w || !targets || !lse || !dz) return FASTMAX_E_NULL;
    if (M <= 0 || V <= 0 || K <= 0 || (K % 64) || (V % 8)) return FASTMAX_E_BAD_SHAPE;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(dz)) & 15) return FASTMAX_E_ALIGNMENT;
    if (((ldx * 2) & 15) || ((ldz * 2) & 15)) return FASTMAX_E_ALIGNMENT;
    GemmParams p{reinterpret_cast<const __bf16*>(x), w, GemmScale{nullptr, nullptr, nullptr, nullptr, 0.f}, nullptr, nullptr, nullptr,
                 reinterpret_cast<__bf16*>(dz), M, V, K, 0, ldx, ldz, (V + 255) / 256, 0, 0, 0, targets, nullptr, nullptr, lse, grad_scale,
                 ignore_index, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0, 0, 0, nullptr};
    return launch_gemm256<false, false, false, false, 2, false, true, true>(p, reinterpret_cast<hipStream_t>(stream));
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
                 (N + 255) / 256, 0, 0, 0, nullptr, nullptr, nullptr, nullptr, 0.f, 0, cos, sin, q, k, v, T, G, q_per_kv, head_size,
                 rope_n_elem, tables16 ? 1 : 0, nullptr};
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
