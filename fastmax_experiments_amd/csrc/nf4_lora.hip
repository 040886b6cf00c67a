// Fused NF4-dequant + LoRA linear for QLoRA fine-tuning on gfx950 (MFMA 16x16x32 bf16).
//
// Replaces, for the frozen 4-bit base layer + low-rank branch of lit_gpt/lora.py
//   LoRALinear.forward      (lora.py:170-177)   y = linear(x) + (dropout(x) A^T) B^T * scaling
//   LoRAQKVLinear.forward   (lora.py:398-433)   y = linear(x) + zero_pad(conv1d(dropout(x) A^T, B)) * scaling
// the bitsandbytes `Linear4bit` matmul the reference reaches through Lightning's BitsandbytesPrecision
// plugin (finetune/lora.py:77; dequantize_4bit call site lora.py:152-161).  bitsandbytes is not part of
// the reference tree: the NF4 format follows the public definition (QLoRA, arXiv 2305.14314: 16-level
// normal-float codebook, block size 64, fp32 absmax per block, two codes per byte, high nibble first);
// parity with bitsandbytes itself is UNPINNED.
//
//   forward   y[M][N]  = x[M][K] . deq(W)[N][K]^T + bias[N] + EA[M][32] . EB[N][32]^T
//             (EA = dropout(x) A^T, EB = scaling * scatter(lora_B): the LoRA branch is ONE extra k-step)
//   backward  dx[M][K] = dy[M][N] . deq(W)[N][K]
//
// Tiling: 128 x 128 output tile per workgroup (4 waves as 2 x 2, 64 x 64 each = 4 x 4 MFMA tiles), BK = 64.
// The weight is the MFMA *A* operand and the activation the *B* operand, so a lane ends up holding 4
// consecutive output columns of one row (8-byte stores).  NF4 codes are expanded in registers through a
// 16-entry LDS table, scaled by the block absmax, packed to bf16 and written to a swizzled LDS tile.
#include "fastmax_common.h"

namespace fastmax {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__constant__ float kNF4[16] = {-1.0f, -0.6961928009986877f, -0.5250730514526367f, -0.39491748809814453f,
                               -0.28444138169288635f, -0.18477343022823334f, -0.09105003625154495f, 0.0f,
                               0.07958029955625534f, 0.16093020141124725f, 0.24611230194568634f,
                               0.33791524171829224f, 0.44070982933044434f, 0.5626170039176941f,
                               0.7229568362236023f, 1.0f};

// Block scales of the NF4 weight: plain (fp32 absmax per 64 weights) or double-quantised ("nf4-dq": the absmax vector is
// itself stored as 8-bit codes of a 256-entry map, one fp32 scale per 256 of them, plus one offset -- QLoRA section 3,
// finetune/lora.py:38 lists the mode).  One scale per 64 weights is decoded where the plain path loads a float.
struct Nf4Scale {
    const float* absmax;      // plain: one per 64 consecutive weights (q == nullptr)
    const uint8_t* q;         // dq: 8-bit code of (absmax - offset) per 64 weights
    const float* absmax2;     // dq: scale per 256 codes
    const float* code2;       // dq: 256-entry map
    float offset;
    __device__ __forceinline__ float operator[](int64_t blk) const {
        return q ? mul_then_add(code2[q[blk]], absmax2[blk >> 8], offset) : absmax[blk];
    }
    // product and sum rounded separately: bit-identical to the host codec (lora.py).  hipcc contracts a * b + c into a
    // fused multiply-add by default, and __fmul_rn / __fadd_rn are plain operators on this target
    static __device__ __forceinline__ float mul_then_add(float a, float b, float c) {
#pragma clang fp contract(off)
        const float prod = a * b;
        return prod + c;
    }
};

struct Nf4Params {
    const void* x;            // fwd: x [M][K]   bwd: dy [M][N]     (bf16 or f32, row-major, ld = ldx)
    const uint8_t* wq;        // packed NF4 codes of W [N][K], row-major, 2 codes / byte
    Nf4Scale absmax;          // block scale of every 64 consecutive weights
    const float* bias;        // [N] or null (fwd only)
    const __bf16* ea;         // [M][32] or null (fwd only)
    const __bf16* eb;         // [N][32] or null
    void* y;                  // fwd: y [M][N]   bwd: dx [M][K]
    int M, N, K;
    int64_t ldx, ldy;
};

__device__ __forceinline__ int sw128(int row, int chunk) { return row * 128 + (((chunk ^ row) & 7) << 4); }

template <typename T> __device__ __forceinline__ bf16x8 load8_as_bf16(const T* p);
template <> __device__ __forceinline__ bf16x8 load8_as_bf16<__bf16>(const __bf16* p) {
    return *reinterpret_cast<const bf16x8*>(p);
}
template <> __device__ __forceinline__ bf16x8 load8_as_bf16<float>(const float* p) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
    bf16x8 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) { o[i] = (__bf16)a[i]; o[4 + i] = (__bf16)b[i]; }
    return o;
}

// 16 bytes of packed codes (32 weights, high nibble first) -> 4 x bf16x8, scaled by `amax`
__device__ __forceinline__ void dequant32(const u32x4 pk, float amax, const float* lut, bf16x8 (&out)[4]) {
#pragma unroll
    for (int wd = 0; wd < 4; ++wd) {
        const unsigned int v = pk[wd];
#pragma unroll
        for (int by = 0; by < 4; ++by) {
            const unsigned int byte = (v >> (8 * by)) & 0xffu;
            out[wd][2 * by] = (__bf16)(lut[byte >> 4] * amax);
            out[wd][2 * by + 1] = (__bf16)(lut[byte & 15u] * amax);
        }
    }
}

template <typename T> __device__ __forceinline__ void store4(T* p, const f32x4 v);
template <> __device__ __forceinline__ void store4<float>(float* p, const f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
template <> __device__ __forceinline__ void store4<__bf16>(__bf16* p, const f32x4 v) {
    bf16x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = (__bf16)v[i];
    *reinterpret_cast<bf16x4*>(p) = o;
}

// ------------------------------------------------------------------------------------------------
// forward: grid = (ceil(N/128), ceil(M/128)), block = 256.  Requires K % 64 == 0, N % 4 == 0.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256, 2) void nf4_linear_fwd_kernel(Nf4Params prm) {
    __shared__ __attribute__((aligned(16))) char smem[2 * 128 * 128 + 64];
    char* Xs = smem;                      // [128 m][64 k] bf16, swizzled 128-byte rows
    char* Ws = smem + 128 * 128;          // [128 n][64 k]
    float* lut = reinterpret_cast<float*>(smem + 2 * 128 * 128);
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = w & 1, wm = w >> 1;
    const int r = lane & 15, q4 = lane >> 4;
    const int n0 = blockIdx.x * 128, m0 = blockIdx.y * 128;
    const int M = prm.M, N = prm.N, K = prm.K;
    const T* X = reinterpret_cast<const T*>(prm.x);
    if (tid < 16) lut[tid] = kNF4[tid];

    // staging maps
    const int xrow = tid >> 3, xchunk = tid & 7;                 // x: rows xrow + 32u, 16-byte chunk (8 elements)
    const int wrow = tid >> 1, whalf = tid & 1;                  // w: one row, 32 codes (16 bytes)
    const int wn_g = min(n0 + wrow, N - 1);
    bf16x8 xr[4];
    u32x4 wr;
    float wa;
    auto issue = [&](int k0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int gm = min(m0 + xrow + 32 * u, M - 1);
            xr[u] = load8_as_bf16<T>(X + (int64_t)gm * prm.ldx + k0 + 8 * xchunk);
        }
        const int64_t e = (int64_t)wn_g * K + k0 + 32 * whalf;   // element index of the first code
        wr = *reinterpret_cast<const u32x4*>(prm.wq + (e >> 1));
        wa = prm.absmax[e >> 6];
    };
    auto stage = [&]() {
#pragma unroll
        for (int u = 0; u < 4; ++u) *reinterpret_cast<bf16x8*>(Xs + sw128(xrow + 32 * u, xchunk)) = xr[u];
        bf16x8 d[4];
        dequant32(wr, wa, lut, d);
#pragma unroll
        for (int c = 0; c < 4; ++c) *reinterpret_cast<bf16x8*>(Ws + sw128(wrow, 4 * whalf + c)) = d[c];
    };

    f32x4 acc[4][4];                                             // [nt][mt]: rows n (regs), col m (lane)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};

    auto compute = [&](int ksteps) {
        for (int ks = 0; ks < ksteps; ++ks) {
            bf16x8 af[4], bfm[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                af[t] = *reinterpret_cast<const bf16x8*>(Ws + sw128(64 * wn + 16 * t + r, 4 * ks + q4));
                bfm[t] = *reinterpret_cast<const bf16x8*>(Xs + sw128(64 * wm + 16 * t + r, 4 * ks + q4));
            }
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[nt], bfm[mt], acc[nt][mt], 0, 0, 0);
        }
    };

    issue(0);
    __syncthreads();                                             // lut visible
    for (int k0 = 0; k0 < K; k0 += 64) {
        stage();
        if (k0 + 64 < K) issue(k0 + 64);                         // next tile in flight under the MFMAs
        __syncthreads();
        compute(2);
        __syncthreads();
    }
    // LoRA branch: one extra k-step over the (padded) rank dimension
    if (prm.ea && prm.eb) {
        if (tid < 256) {
            const int row = tid >> 1, c2 = tid & 1;                  // 128 rows x 64 bytes = 2 x (2 chunks) per row
            const int gm = min(m0 + row, M - 1), gn = min(n0 + row, N - 1);
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                *reinterpret_cast<bf16x8*>(Xs + sw128(row, 2 * c2 + c)) =
                    *reinterpret_cast<const bf16x8*>(prm.ea + (int64_t)gm * 32 + 8 * (2 * c2 + c));
                *reinterpret_cast<bf16x8*>(Ws + sw128(row, 2 * c2 + c)) =
                    *reinterpret_cast<const bf16x8*>(prm.eb + (int64_t)gn * 32 + 8 * (2 * c2 + c));
            }
        }
        __syncthreads();
        compute(1);
    }
    // epilogue: lane holds y[m = .. + r][n = .. + 4*q4 + 0..3]
    T* Y = reinterpret_cast<T*>(prm.y);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int gm = m0 + 64 * wm + 16 * mt + r;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int gn = n0 + 64 * wn + 16 * nt + 4 * q4;
            if (gm < M && gn < N) {
                f32x4 v = acc[nt][mt];
                if (prm.bias) v += *reinterpret_cast<const f32x4*>(prm.bias + gn);
                store4<T>(Y + (int64_t)gm * prm.ldy + gn, v);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// backward wrt the input: dx[M][K] = dy[M][N] . deq(W)[N][K].  Contraction over n.
// grid = (K/128, ceil(M/128)), block = 256.  Requires K % 128 == 0, N % 64 == 0.
// W tile [64 n][128 k] bf16 with 256-byte rows; A operand (rows k) by transposed reads.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int sw256(int row, int chunk) { return row * 256 + (((chunk ^ (2 * row)) & 15) << 4); }

template <typename T>
__global__ __launch_bounds__(256, 2) void nf4_linear_dx_kernel(Nf4Params prm) {
    __shared__ __attribute__((aligned(16))) char smem[2 * 128 * 128 + 64];
    char* Gs = smem;                      // dy tile [128 m][64 n] bf16, 128-byte rows
    char* Ws = smem + 128 * 128;          // W tile  [64 n][128 k] bf16, 256-byte rows
    float* lut = reinterpret_cast<float*>(smem + 2 * 128 * 128);
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wk = w & 1, wm = w >> 1;
    const int r = lane & 15, q4 = lane >> 4;
    const int kk0 = blockIdx.x * 128, m0 = blockIdx.y * 128;
    const int M = prm.M, N = prm.N, K = prm.K;
    const T* G = reinterpret_cast<const T*>(prm.x);
    if (tid < 16) lut[tid] = kNF4[tid];

    const int grow = tid >> 3, gchunk = tid & 7;                 // dy: rows grow + 32u, 8 elements
    const int wrow = tid >> 2, wpiece = tid & 3;                 // W: row n, 32 codes at k = kk0 + 32*wpiece
    bf16x8 gr[4];
    u32x4 wr;
    float wa;
    auto issue = [&](int nn0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int gm = min(m0 + grow + 32 * u, M - 1);
            gr[u] = load8_as_bf16<T>(G + (int64_t)gm * prm.ldx + nn0 + 8 * gchunk);
        }
        const int64_t e = (int64_t)(nn0 + wrow) * K + kk0 + 32 * wpiece;
        wr = *reinterpret_cast<const u32x4*>(prm.wq + (e >> 1));
        wa = prm.absmax[e >> 6];
    };
    auto stage = [&]() {
#pragma unroll
        for (int u = 0; u < 4; ++u) *reinterpret_cast<bf16x8*>(Gs + sw128(grow + 32 * u, gchunk)) = gr[u];
        bf16x8 d[4];
        dequant32(wr, wa, lut, d);
#pragma unroll
        for (int c = 0; c < 4; ++c) *reinterpret_cast<bf16x8*>(Ws + sw256(wrow, 4 * wpiece + c)) = d[c];
    };
    f32x4 acc[4][4];                                             // [kt][mt]: rows k (regs), col m (lane)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};

    const int qq = (lane & 15) >> 2, pp = lane & 3;
    issue(0);
    __syncthreads();
    for (int nn0 = 0; nn0 < N; nn0 += 64) {
        stage();
        if (nn0 + 64 < N) issue(nn0 + 64);
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {                         // 32 values of n per step
            bf16x8 af[4], bfm[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                // A[row k][n]: transposed read, rows n = 32ks + 4q4 + qq (+16), columns k = 64wk + 16t + 4pp..
                const int col = 64 * wk + 16 * t;
                const int ra = 32 * ks + 4 * q4 + qq, rb = ra + 16;
                const int chunk = (col >> 3) + (pp >> 1), half = (pp & 1) << 3;
                union { bf16x8 v; s16x4 h[2]; } u;
                u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4*)(Ws + sw256(ra, chunk) + half));
                u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4*)(Ws + sw256(rb, chunk) + half));
                af[t] = u.v;
                // B[n][col m]: dy row m, the same permuted n order: n = 32ks + 4q4 + {0..3}, +16
                const int mrow = 64 * wm + 16 * t + r;
                const int e0 = 32 * ks + 4 * q4;                 // element offset inside the 64-wide row
                union { bf16x8 v; bf16x4 h[2]; } b;
                b.h[0] = *reinterpret_cast<const bf16x4*>(Gs + sw128(mrow, e0 >> 3) + ((e0 & 7) << 1));
                b.h[1] = *reinterpret_cast<const bf16x4*>(Gs + sw128(mrow, (e0 + 16) >> 3) + (((e0 + 16) & 7) << 1));
                bfm[t] = b.v;
            }
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    acc[kt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[kt], bfm[mt], acc[kt][mt], 0, 0, 0);
        }
        __syncthreads();
    }
    T* DX = reinterpret_cast<T*>(prm.y);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int gm = m0 + 64 * wm + 16 * mt + r;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            const int gk = kk0 + 64 * wk + 16 * kt + 4 * q4;
            if (gm < M) store4<T>(DX + (int64_t)gm * prm.ldy + gk, acc[kt][mt]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// forward for a handful of rows (M <= 16: generation, one token per sequence): the 128 x 128 tile kernel would put
// N/128 workgroups on the chip.  Here a workgroup owns 16 output columns and its four waves split K in 128-wide blocks
// (block j -> wave j % 4); per block a lane loads 16 bytes of codes (32 weights of ONE row: A operand rows = output
// columns) and the matching 64 bytes of x (B operand), decodes in registers and issues four MFMA 16x16x32.  The four
// partial 16 x 16 tiles meet in LDS; wave 0 adds the LoRA k-step and the bias.  HBM-bound on the 0.5 byte / weight codes.
// grid = N/16, block = 256.  Requires K % 128 == 0, N % 16 == 0, M <= 16.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void nf4_gemv_kernel(Nf4Params prm) {
    __shared__ float lut[16];
    __shared__ __attribute__((aligned(16))) float part[4][16 * 16];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q4 = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int M = prm.M, N = prm.N, K = prm.K;
    const T* X = reinterpret_cast<const T*>(prm.x);
    if (tid < 16) lut[tid] = kNF4[tid];
    __syncthreads();
    const int n = n0 + r;
    const bool mrow = r < M;
    const T* xrow = X + (int64_t)(mrow ? r : 0) * prm.ldx;
    f32x4 acc = {0, 0, 0, 0};
    const int nblk = K / 128;
    // two 128-wide blocks per trip: both blocks' codes, scales and x pieces are requested before either is decoded
    auto fetch = [&](int j, u32x4& pk, float& amax, bf16x8 (&xf)[4]) {
        const int64_t e = (int64_t)n * K + 128 * j + 32 * q4;             // first of this lane's 32 weights
        pk = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(prm.wq + (e >> 1)));
        amax = prm.absmax[e >> 6];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (mrow) xf[t] = load8_as_bf16<T>(xrow + 128 * j + 32 * q4 + 8 * t);
            else
#pragma unroll
                for (int i = 0; i < 8; ++i) xf[t][i] = (__bf16)0.0f;
        }
    };
    auto consume = [&](const u32x4& pk, float amax, const bf16x8 (&xf)[4]) {
        bf16x8 wf[4];
        dequant32(pk, amax, lut, wf);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[t], xf[t], acc, 0, 0, 0);
    };
    int j = w;
    for (; j + 4 < nblk; j += 8) {
        u32x4 pk0, pk1;
        float a0, a1;
        bf16x8 x0[4], x1[4];
        fetch(j, pk0, a0, x0);
        fetch(j + 4, pk1, a1, x1);
        consume(pk0, a0, x0);
        consume(pk1, a1, x1);
    }
    if (j < nblk) {
        u32x4 pk0;
        float a0;
        bf16x8 x0[4];
        fetch(j, pk0, a0, x0);
        consume(pk0, a0, x0);
    }
    // C layout: column (lane & 15) = row m of x, rows 4 q4 + i = output column n0 + 4 q4 + i
#pragma unroll
    for (int i = 0; i < 4; ++i) part[w][(4 * q4 + i) * 16 + r] = acc[i];
    __syncthreads();
    if (w == 0) {
        f32x4 tot;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            tot[i] = (part[0][(4 * q4 + i) * 16 + r] + part[1][(4 * q4 + i) * 16 + r]) + (part[2][(4 * q4 + i) * 16 + r] + part[3][(4 * q4 + i) * 16 + r]);
        if (prm.ea) {
            bf16x8 af = *reinterpret_cast<const bf16x8*>(prm.eb + (int64_t)n * 32 + 8 * q4), bf;
            if (mrow) bf = *reinterpret_cast<const bf16x8*>(prm.ea + (int64_t)r * 32 + 8 * q4);
            else
#pragma unroll
                for (int i = 0; i < 8; ++i) bf[i] = (__bf16)0.0f;
            tot = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf, tot, 0, 0, 0);
        }
        if (prm.bias) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(prm.bias + n0 + 4 * q4);
            tot += b4;
        }
        if (mrow) store4<T>(reinterpret_cast<T*>(prm.y) + (int64_t)r * prm.ldy + n0 + 4 * q4, tot);
    }
}

// ---- dequantise to a dense matrix: the merge path (lora.py:142-168) and the large-M route of the QLoRA linear, where
// the W tile would otherwise be re-decoded by every one of the M/128 workgroup rows: decode once, then a plain GEMM.
__global__ __launch_bounds__(256) void nf4_dequant_bf16_vec_kernel(const uint8_t* wq, const Nf4Scale absmax, __bf16* out, int64_t n8) {
    // one thread = 4 packed bytes = 8 weights = one 16-byte store: consecutive lanes read consecutive words and write
    // consecutive 16-byte pieces (the earlier 32-weights-per-thread form wrote 64-byte-strided pieces: 40 us for 16.7 M weights)
    __shared__ float lut[16];
    if (threadIdx.x < 16) lut[threadIdx.x] = kNF4[threadIdx.x];
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;                  // index of the 8-weight piece
    if (i >= n8) return;
    const unsigned int word = __builtin_nontemporal_load(reinterpret_cast<const unsigned int*>(wq) + i);
    const float a = absmax[i >> 3];
    typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
    bf16x8_t o;
#pragma unroll
    for (int by = 0; by < 4; ++by) {
        const unsigned int byte = (word >> (8 * by)) & 0xffu;                // bytes in memory order = bits 0..7 first
        o[2 * by] = (__bf16)(lut[byte >> 4] * a);                            // high nibble first (bitsandbytes order)
        o[2 * by + 1] = (__bf16)(lut[byte & 15u] * a);
    }
    __builtin_nontemporal_store(o, reinterpret_cast<bf16x8_t*>(out) + i);
}
template <typename T>
__global__ void nf4_dequant_kernel(const uint8_t* wq, const Nf4Scale absmax, T* out, int64_t n) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2;
    if (i >= n) return;
    const unsigned int byte = wq[i >> 1];
    const float a = absmax[i >> 6];
    out[i] = (T)(kNF4[byte >> 4] * a);
    if (i + 1 < n) out[i + 1] = (T)(kNF4[byte & 15u] * a);
}

}  // namespace fastmax

using namespace fastmax;

extern "C" {

static Nf4Scale to_scale(const fastmax_nf4_scales* sc) {
    return Nf4Scale{sc->absmax, sc->absmax_q, sc->absmax2, sc->code2, sc->offset};
}
static int check_scale(const fastmax_nf4_scales* sc) {
    if (!sc) return FASTMAX_E_NULL;
    if (sc->absmax_q) return (sc->absmax2 && sc->code2) ? FASTMAX_OK : FASTMAX_E_NULL;
    return sc->absmax ? FASTMAX_OK : FASTMAX_E_NULL;
}

int fastmax_hip_nf4_linear_forward_s(const void* x, int64_t ldx, const uint8_t* wq, const fastmax_nf4_scales* scales,
                                     const float* bias, const void* ea, const void* eb, void* y, int64_t ldy, int M, int N,
                                     int K, int dtype, void* stream) {
    if (!x || !wq || !y) return FASTMAX_E_NULL;
    if (int rc = check_scale(scales)) return rc;
    if (M <= 0 || N <= 0 || K <= 0 || (K % 64) || (N % 4)) return FASTMAX_E_BAD_SHAPE;
    if ((ea == nullptr) != (eb == nullptr)) return FASTMAX_E_NULL;
    const int es = dtype == FASTMAX_F32 ? 4 : 2;
    if ((reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(y) & 15) || ((ldx * es) & 15) ||
        ((ldy * es) & 15) || (reinterpret_cast<uintptr_t>(wq) & 15))
        return FASTMAX_E_ALIGNMENT;
    Nf4Params p{x, wq, to_scale(scales), bias, reinterpret_cast<const __bf16*>(ea), reinterpret_cast<const __bf16*>(eb), y, M, N, K,
                ldx, ldy};
    dim3 grid((N + 127) / 128, (M + 127) / 128), block(256);
    if (M <= 16 && (K % 128) == 0 && (N % 16) == 0 && (!bias || !(reinterpret_cast<uintptr_t>(bias) & 15)) &&
        (!ea || !((reinterpret_cast<uintptr_t>(ea) | reinterpret_cast<uintptr_t>(eb)) & 15))) {
        // generation-size row counts: column-sliced kernel that fills the chip (N/16 workgroups)
        if (dtype == FASTMAX_BF16) hipLaunchKernelGGL(nf4_gemv_kernel<__bf16>, dim3(N / 16), block, 0, (hipStream_t)stream, p);
        else if (dtype == FASTMAX_F32) hipLaunchKernelGGL(nf4_gemv_kernel<float>, dim3(N / 16), block, 0, (hipStream_t)stream, p);
        else return FASTMAX_E_BAD_DTYPE;
        return (int)hipGetLastError();
    }
    if (dtype == FASTMAX_BF16) hipLaunchKernelGGL(nf4_linear_fwd_kernel<__bf16>, grid, block, 0, (hipStream_t)stream, p);
    else if (dtype == FASTMAX_F32) hipLaunchKernelGGL(nf4_linear_fwd_kernel<float>, grid, block, 0, (hipStream_t)stream, p);
    else return FASTMAX_E_BAD_DTYPE;
    return (int)hipGetLastError();
}

int fastmax_hip_nf4_linear_backward_input_s(const void* dy, int64_t lddy, const uint8_t* wq, const fastmax_nf4_scales* scales,
                                            void* dx, int64_t lddx, int M, int N, int K, int dtype, void* stream) {
    if (!dy || !wq || !dx) return FASTMAX_E_NULL;
    if (int rc = check_scale(scales)) return rc;
    if (M <= 0 || N <= 0 || K <= 0 || (K % 128) || (N % 64)) return FASTMAX_E_BAD_SHAPE;
    const int es = dtype == FASTMAX_F32 ? 4 : 2;
    if ((reinterpret_cast<uintptr_t>(dy) & 15) || (reinterpret_cast<uintptr_t>(dx) & 15) || ((lddy * es) & 15) ||
        ((lddx * es) & 15) || (reinterpret_cast<uintptr_t>(wq) & 15))
        return FASTMAX_E_ALIGNMENT;
    Nf4Params p{dy, wq, to_scale(scales), nullptr, nullptr, nullptr, dx, M, N, K, lddy, lddx};
    dim3 grid(K / 128, (M + 127) / 128), block(256);
    if (dtype == FASTMAX_BF16) hipLaunchKernelGGL(nf4_linear_dx_kernel<__bf16>, grid, block, 0, (hipStream_t)stream, p);
    else if (dtype == FASTMAX_F32) hipLaunchKernelGGL(nf4_linear_dx_kernel<float>, grid, block, 0, (hipStream_t)stream, p);
    else return FASTMAX_E_BAD_DTYPE;
    return (int)hipGetLastError();
}

int fastmax_hip_nf4_dequantize_s(const uint8_t* wq, const fastmax_nf4_scales* scales, void* out, int64_t n, int dtype, void* stream) {
    if (!wq || !out) return FASTMAX_E_NULL;
    if (int rc = check_scale(scales)) return rc;
    if (n <= 0 || (n % 64)) return FASTMAX_E_BAD_SHAPE;
    const Nf4Scale sc = to_scale(scales);
    const int64_t threads = (n + 1) / 2;
    dim3 grid((unsigned)((threads + 255) / 256)), block(256);
    if (dtype == FASTMAX_BF16 && !((reinterpret_cast<uintptr_t>(wq) | reinterpret_cast<uintptr_t>(out)) & 15)) {
        const int64_t n8 = n / 8;
        hipLaunchKernelGGL(nf4_dequant_bf16_vec_kernel, dim3((unsigned)((n8 + 255) / 256)), block, 0, (hipStream_t)stream, wq,
                           sc, (__bf16*)out, n8);
    } else if (dtype == FASTMAX_BF16)
        hipLaunchKernelGGL(nf4_dequant_kernel<__bf16>, grid, block, 0, (hipStream_t)stream, wq, sc, (__bf16*)out, n);
    else if (dtype == FASTMAX_F32)
        hipLaunchKernelGGL(nf4_dequant_kernel<float>, grid, block, 0, (hipStream_t)stream, wq, sc, (float*)out, n);
    else return FASTMAX_E_BAD_DTYPE;
    return (int)hipGetLastError();
}

// plain NF4 (fp32 block scales): the same entry points with the scales given as one pointer
int fastmax_hip_nf4_linear_forward(const void* x, int64_t ldx, const uint8_t* wq, const float* absmax, const float* bias,
                                   const void* ea, const void* eb, void* y, int64_t ldy, int M, int N, int K, int dtype,
                                   void* stream) {
    const fastmax_nf4_scales sc{absmax, nullptr, nullptr, nullptr, 0.f};
    return fastmax_hip_nf4_linear_forward_s(x, ldx, wq, &sc, bias, ea, eb, y, ldy, M, N, K, dtype, stream);
}
int fastmax_hip_nf4_linear_backward_input(const void* dy, int64_t lddy, const uint8_t* wq, const float* absmax, void* dx,
                                          int64_t lddx, int M, int N, int K, int dtype, void* stream) {
    const fastmax_nf4_scales sc{absmax, nullptr, nullptr, nullptr, 0.f};
    return fastmax_hip_nf4_linear_backward_input_s(dy, lddy, wq, &sc, dx, lddx, M, N, K, dtype, stream);
}
int fastmax_hip_nf4_dequantize(const uint8_t* wq, const float* absmax, void* out, int64_t n, int dtype, void* stream) {
    const fastmax_nf4_scales sc{absmax, nullptr, nullptr, nullptr, 0.f};
    return fastmax_hip_nf4_dequantize_s(wq, &sc, out, n, dtype, stream);
}

}  // extern "C"
