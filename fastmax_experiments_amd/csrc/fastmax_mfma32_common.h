// Shared pieces of the 32x32x16 bf16 MFMA kernels (gfx950): operand fragments for 32-row tiles and the transposed
// accumulator store.  Layouts: A row = lane&31, k = 8(lane>>5)+j; B col = lane&31, same k;
// C col = lane&31, row(i) = (i&3) + 8(i>>2) + 4(lane>>5).
#pragma once
#include "fastmax_mfma_common.h"

namespace fastmax {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ f32x16 mfma32(const bf16x8 a, const bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// first product of an accumulation chain that starts at the constant 1.0: the compiler folds an inline-constant C
// operand only when the constant has a single use, so the instruction is written out.  hipcc pads nothing around an asm
// statement: `s_nop 1` covers a vector-ALU write of an operand just before it (e.g. an accumulator-file read), and the
// trailing `s_nop 11` (12 wait states, an 8-pass result) covers ANY reader of the result -- the next MFMA of the chain
// would need none, but in a kernel whose accumulators live in the AGPR half of the file the compiler moves the result
// with v_accvgpr_write, or spills it, without knowing a matrix instruction produced it.  The pad is free in a chain: the
// dependent MFMA cannot start before this one has left the pipe (32 cycles).
__device__ __forceinline__ f32x16 mfma32_c1(const bf16x8 a, const bf16x8 b) {
    f32x16 d;
    asm("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, 1.0\n\ts_nop 11" : "=&v"(d) : "v"(a), "v"(b));
    return d;
}
template <int NA, int NB>
__device__ __forceinline__ f32x16 mfma32_parts_c1(const Frag<NA>& a, const Frag<NB>& b) {
    f32x16 c = mfma32_c1(a.p[0], b.p[0]);
    if constexpr (NA == 2) c = mfma32(a.p[1], b.p[0], c);
    if constexpr (NB == 2) c = mfma32(a.p[0], b.p[1], c);
    return c;
}
template <int NA, int NB>
__device__ __forceinline__ f32x16 mfma32_parts(const Frag<NA>& a, const Frag<NB>& b, f32x16 c) {
    c = mfma32(a.p[0], b.p[0], c);
    if constexpr (NA == 2) c = mfma32(a.p[1], b.p[0], c);
    if constexpr (NB == 2) c = mfma32(a.p[0], b.p[1], c);
    return c;
}

// Accumulate into a tile that lives in the ACCUMULATOR half of the register file (AGPRs).  The one-wave-per-SIMD backward
// kernels at D = 128 keep their output accumulators (64 - 128 registers that only matrix instructions touch) there, while the
// score chains -- whose results the vector ALU reads -- are compiled in the VGPR form (-amdgpu-mfma-vgpr-form): the compiler
// selects one form for every MFMA of a kernel, so the accumulator-file ones are written out.  `s_nop 1`: a vector-ALU write of
// an operand just before the statement (the packed score tile).  Chains into the same accumulator need no wait states; any
// other reader must come after acc_fence().
__device__ __forceinline__ void mfma32_acc(f32x16& acc, const bf16x8 a, const bf16x8 b) {
    asm("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
template <int NA, int NB>
__device__ __forceinline__ void mfma32_parts_acc(f32x16& acc, const Frag<NA>& a, const Frag<NB>& b) {
    mfma32_acc(acc, a.p[0], b.p[0]);
    if constexpr (NA == 2) mfma32_acc(acc, a.p[1], b.p[0]);
    if constexpr (NB == 2) mfma32_acc(acc, a.p[0], b.p[1]);
}
// 16 wait states between the last matrix instruction into `acc` and whatever the compiler does with it next
__device__ __forceinline__ void acc_fence(f32x16& acc) { asm volatile("s_nop 15" : "+a"(acc)); }

// A operand of the 32x32x16 MFMA from a row-major image, transposed: lane (d = lane&31, h = lane>>5) receives
// rows row0 + 4h + {0..3} and row0 + 8 + 4h + {0..3} of image column col0 + d  (= the key order of the accumulator
// registers 8s..8s+7 of an S^T tile).
template <int DP, int SW = 2> __device__ __forceinline__ bf16x8 ld_tr8_32(const char* smem, int base, int row0, int col0, int lane) {
    const int g = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;
    const int ra = row0 + 4 * (g >> 1) + qq, rb = ra + 8;
    const int col = col0 + 16 * (g & 1) + 4 * pp;
    const int chunk = col >> 3, half = (col & 4) << 1;
    union { bf16x8 v; s16x4 h[2]; } u;
    u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)(smem + base + img_off<DP, SW>(ra, chunk) + half));
    u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)(smem + base + img_off<DP, SW>(rb, chunk) + half));
    return u.v;
}

// 8 consecutive elements (columns col0..col0+7) of one query row -> B fragment part(s); zero outside the tensor
template <typename TIN>
__device__ __forceinline__ Frag<InTraits<TIN>::NP> load_q_frag(const TIN* base, int64_t sn, int row, int nrows, int col0, int D) {
    constexpr int EPL = InTraits<TIN>::EPL, NP = InTraits<TIN>::NP;
    Frag<NP> f;
    if constexpr (NP == 1) {
        f.p[0] = __builtin_bit_cast(bf16x8, load_piece<TIN>(base, sn, row, nrows, col0 / EPL, D));
    } else {
        float x[8];
        if constexpr (EPL == 4) {
            float lo4[4], hi4[4];
            piece_to_float<TIN>(load_piece<TIN>(base, sn, row, nrows, col0 / 4, D), lo4);
            piece_to_float<TIN>(load_piece<TIN>(base, sn, row, nrows, col0 / 4 + 1, D), hi4);
#pragma unroll
            for (int i = 0; i < 4; ++i) { x[i] = lo4[i]; x[4 + i] = hi4[i]; }
        } else {
            piece_to_float<TIN>(load_piece<TIN>(base, sn, row, nrows, col0 / 8, D), x);
        }
        bf16x4 h0, l0, h1, l1;
        split4(f32x4{x[0], x[1], x[2], x[3]}, h0, l0);
        split4(f32x4{x[4], x[5], x[6], x[7]}, h1, l1);
        f.p[0] = cat4(h0, h1);
        f.p[1] = cat4(l0, l1);
    }
    return f;
}


// The same fragment with every element multiplied by `scale` first.  Exact for bf16 data when scale is a power of two
// (the caller's condition for using it on single-part operands); split operands are scaled in fp32 before the split.
template <typename TIN>
__device__ __forceinline__ Frag<InTraits<TIN>::NP> load_q_frag_scaled(const TIN* base, int64_t sn, int row, int nrows, int col0, int D, float scale) {
    constexpr int EPL = InTraits<TIN>::EPL, NP = InTraits<TIN>::NP;
    float x[8];
    if constexpr (EPL == 4) {
        float lo4[4], hi4[4];
        piece_to_float<TIN>(load_piece<TIN>(base, sn, row, nrows, col0 / 4, D), lo4);
        piece_to_float<TIN>(load_piece<TIN>(base, sn, row, nrows, col0 / 4 + 1, D), hi4);
#pragma unroll
        for (int i = 0; i < 4; ++i) { x[i] = lo4[i]; x[4 + i] = hi4[i]; }
    } else {
        piece_to_float<TIN>(load_piece<TIN>(base, sn, row, nrows, col0 / 8, D), x);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] *= scale;
    Frag<NP> f;
    if constexpr (NP == 1) {
        f.p[0] = cat4(to_bf16x4(f32x4{x[0], x[1], x[2], x[3]}), to_bf16x4(f32x4{x[4], x[5], x[6], x[7]}));
    } else {
        bf16x4 h0, l0, h1, l1;
        split4(f32x4{x[0], x[1], x[2], x[3]}, h0, l0);
        split4(f32x4{x[4], x[5], x[6], x[7]}, h1, l1);
        f.p[0] = cat4(h0, h1);
        f.p[1] = cat4(l0, l1);
    }
    return f;
}

// Write a wave's transposed accumulator tiles (acc[dt][i]: column 32dt + row(i) of tensor row first_row + (lane&31))
// times `scale` (per lane = per tensor row) as whole row segments, 32 rows x 32 columns at a time through a
// wave-private 4 KiB LDS area.  row0_elem = element index of tensor row 0 of this (b,h) slab / D.
template <int DT>
__device__ __forceinline__ void store_tile32_t(char* area, const f32x16 (&acc)[DT], float scale, int lane, void* out, int dtype,
                                               int64_t slab_row0, int first_row, int nrows, int D) {
    const int l31 = lane & 31, h = lane >> 5;
    const bool out32 = dtype == FASTMAX_F32;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
        for (int ig = 0; ig < 4; ++ig) {
            const f32x4 val = f32x4{acc[dt][4 * ig], acc[dt][4 * ig + 1], acc[dt][4 * ig + 2], acc[dt][4 * ig + 3]} * scale;
            if (out32) {
                *reinterpret_cast<f32x4*>(area + l31 * 128 + ((((2 * ig + h) ^ l31) & 7) << 4)) = val;
            } else {
                char* dst = area + l31 * 64 + (((ig ^ l31) & 3) << 4) + (h << 3);
                if (dtype == FASTMAX_BF16) *reinterpret_cast<bf16x4*>(dst) = to_bf16x4(val);
                else {
                    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                    h4 o;
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[i] = (_Float16)val[i];
                    *reinterpret_cast<h4*>(dst) = o;
                }
            }
        }
        if (out32) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = u * 64 + lane, rl = idx >> 3, cc = idx & 7;
                const f32x4 val = *reinterpret_cast<const f32x4*>(area + rl * 128 + (((cc ^ rl) & 7) << 4));
                const int col = 32 * dt + 4 * cc;
                if (first_row + rl < nrows && col < D)
                    __builtin_nontemporal_store(val, reinterpret_cast<f32x4*>(reinterpret_cast<float*>(out) + (slab_row0 + first_row + rl) * D + col));
            }
        } else {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int idx = u * 64 + lane, rl = idx >> 2, cc = idx & 3;
                const u32x4 val = *reinterpret_cast<const u32x4*>(area + rl * 64 + (((cc ^ rl) & 3) << 4));
                const int col = 32 * dt + 8 * cc;
                if (first_row + rl < nrows && col < D)
                    __builtin_nontemporal_store(val, reinterpret_cast<u32x4*>(reinterpret_cast<uint16_t*>(out) + (slab_row0 + first_row + rl) * D + col));
            }
        }
    }
}

}  // namespace fastmax
