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
};

__constant__ float kGemmNF4[16] = {-1.0f, -0.6961928009986877f, -0.5250730514526367f, -0.39491748809814453f,
                                   -0.28444138169288635f, -0.18477343022823334f, -0.09105003625154495f, 0.0f,
                                   0.07958029955625534f, 0.16093020141124725f, 0.24611230194568634f,
                                   0.33791524171829224f, 0.44070982933044434f, 0.5626170039176941f,
                                   0.7229568362236023f, 1.0f};

__device__ __forceinline__ int gsw(int row, int chunk) { return row * 128 + (((chunk ^ row) & 7) << 4); }

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
template <bool WNF4, bool HALVES, bool PF = false, bool ILV = false>
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
    const int bn = blockIdx.x % prm.nbn, bm = blockIdx.x / prm.nbn;
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
    // PF: the line this thread touches in every tile (threads 0..255: x row, 256..511: W row)
    unsigned int pf_sink = 0;
    const char* pf_base = nullptr;
    if constexpr (PF) {
        const int prow = tid & 255;
        pf_base = tid < 256 ? reinterpret_cast<const char*>(prm.x + (int64_t)min(m0 + prow, M - 1) * prm.ldx)
                            : reinterpret_cast<const char*>(reinterpret_cast<const __bf16*>(prm.w) + (int64_t)min(n0 + prow, N - 1) * K);
    }
    if constexpr (ILV) {
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
        } else {
            __syncthreads();                                         // tile kt+1 landed (DMA drained by the barrier's wait), tile kt consumed
        }
    }
    if constexpr (PF) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("" ::"v"(pf_sink));
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

template <bool WNF4, bool HALVES, bool PF = false, bool ILV = false>
static int launch_gemm256(const GemmParams& p, hipStream_t stream) {
    auto kern = qlora_gemm256_kernel<WNF4, HALVES, PF, ILV>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, g256::LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const int nbm = (p.M + 255) / 256;
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
                 reinterpret_cast<const __bf16*>(eb), reinterpret_cast<__bf16*>(y), M, N, K, rank_pad, ldx, ldy, (N + 255) / 256};
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    // dense weight: "gemm_sched" 5 = the L2-prefetch form (measured 5-9 % slower than the plain two-stage loop: kept for A/B)
    const int sched = tune_get(TUNE_GEMM_SCHED);
    if (!w_is_nf4) {
        if (sched == 5) return launch_gemm256<false, false, true>(p, st);
        return sched == 6 ? launch_gemm256<false, false, false, true>(p, st) : launch_gemm256<false, false, false>(p, st);
    }
    if (sched == 6) return launch_gemm256<true, false, false, true>(p, st);
    // NF4 in the loop: "gemm_sched" 1 = SIMD partner waves decode / multiply in opposite order (measured 5-10 % slower), else
    // every wave decodes after its matrix instructions
    return tune_get(TUNE_GEMM_SCHED) == 1 ? launch_gemm256<true, true>(p, st) : launch_gemm256<true, false>(p, st);
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
