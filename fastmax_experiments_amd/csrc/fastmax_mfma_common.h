// Shared MFMA / LDS-image helpers for the fastmax matrix-core kernels (gfx950).
#pragma once
#include "fastmax_common.h"

#include <type_traits>

namespace fastmax {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Byte offset of 16-byte chunk `chunk` of row `row` in a swizzled row-major bf16 image with DP columns.
//   DP = 64 : 128-byte rows, chunk ^ (row & 7)
//   DP = 128: 256-byte rows, chunk ^ 2*(row & 7)
// Both are conflict-free for ds_read_b128 row reads (lane = row) and for ds_read_b64_tr_b16 blocks of
// 4 rows x 16 columns taken at rows 4q+{0..3} / 16+4q+{0..3} (see tools/mfma_probe.hip).
// 64-row bf16 tile images in LDS.
// SW = 0: XOR swizzle for 16-row fragments (16x16x32 MFMA operands), dense rows.
// SW = 1 / 2: padded rows for 32-row fragments (32x32x16 operands), no swizzle, so every fragment address is one lane
// constant plus an immediate:
//   1 (row reads, ds_read_b128): row stride 2 DP + 16 bytes -- the lane groups {0-3,12-15,20-27} / {4-11,16-19,28-31}
//     of one chunk column land on 16 distinct 16-byte slots
//   2 (transposed reads, ds_read_b64_tr_b16): row stride 2 DP + 64 bytes -- the 4 consecutive rows x 64 bytes of a
//     half-wave land on the four 64-byte quarters of the 256-byte bank row
//   3: dense rows with an XOR key that serves BOTH kinds of read of 32-row fragments (an image that is read by rows for
//     one product and transposed for another): key = (row bit 1, row bits 4:3) at 128-byte rows and
//     (row bits 1:0, row bits 4:3) at 256-byte rows; costs one address register per (k-step, chunk block) variant
template <int DP, int SW = 0> constexpr int img_row_bytes() { return (SW == 0 || SW == 3) ? 2 * DP : (SW == 1 ? 2 * DP + 16 : 2 * DP + 64); }
template <int DP, int SW = 0> constexpr int img_bytes() { return 64 * img_row_bytes<DP, SW>(); }
template <int DP, int SW = 0> __device__ __forceinline__ int img_off(int row, int chunk) {
    if constexpr (SW == 0) {
        if constexpr (DP == 64) return row * 128 + (((chunk ^ row) & 7) << 4);
        else if constexpr (DP == 128) return row * 256 + (((chunk ^ (2 * (row & 7))) & 15) << 4);
        else return row * (2 * DP) + (((chunk ^ (row & 15)) & (DP / 8 - 1)) << 4);     // 512-byte rows (head sizes up to 256)
    } else if constexpr (SW == 3) {
        if constexpr (DP == 64) return row * 128 + (((chunk ^ ((((row >> 1) & 1) << 2) | ((row >> 3) & 3))) & 7) << 4);
        else return row * 256 + (((chunk ^ (((row & 3) << 2) | ((row >> 3) & 3))) & 15) << 4);
    } else {
        return row * img_row_bytes<DP, SW>() + (chunk << 4);
    }
}

__device__ __forceinline__ void split4(const f32x4 x, bf16x4& hi, bf16x4& lo) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        hi[i] = (__bf16)x[i];
        lo[i] = (__bf16)(x[i] - (float)hi[i]);
    }
}
__device__ __forceinline__ bf16x4 to_bf16x4(const f32x4 x) {
    bf16x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = (__bf16)x[i];
    return o;
}
__device__ __forceinline__ bf16x8 cat4(const bf16x4 a, const bf16x4 b) {
    return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <int DP, int SW = 0> __device__ __forceinline__ bf16x8 ld_row8(const char* smem, int base, int row, int chunk) {
    return *reinterpret_cast<const bf16x8*>(smem + base + img_off<DP, SW>(row, chunk));
}

// Row fragment in the PERMUTED k order (elements 0..3 = columns col0+4q.., 4..7 = col0+16+4q..): the
// partner of an operand that comes from accumulator tiles or from a transposed read.
template <int DP> __device__ __forceinline__ bf16x8 ld_row8_perm(const char* smem, int base, int row, int col0, int q) {
    const int e0 = col0 + 4 * q, e1 = e0 + 16;
    const bf16x4 a = *reinterpret_cast<const bf16x4*>(smem + base + img_off<DP>(row, e0 >> 3) + ((e0 & 7) << 1));
    const bf16x4 b = *reinterpret_cast<const bf16x4*>(smem + base + img_off<DP>(row, e1 >> 3) + ((e1 & 7) << 1));
    return cat4(a, b);
}

// Transposed fragment: lane (r = lane&15, q = lane>>4) receives, for image column col0 + r, the 8 rows
// row0 + 4q + {0..3} and row0 + 16 + 4q + {0..3}.  Address lanes: lane 4q'+p' of a 16-lane group supplies
// row q' of the 4-row block, columns 4p'..4p'+3.
template <int DP> __device__ __forceinline__ bf16x8 ld_tr8(const char* smem, int base, int row0, int col0, int lane) {
    const int q = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;
    const int ra = row0 + 4 * q + qq, rb = ra + 16;
    const int chunk = (col0 >> 3) + (pp >> 1), half = (pp & 1) << 3;
    union { bf16x8 v; s16x4 h[2]; } u;
    u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)(smem + base + img_off<DP>(ra, chunk) + half));
    u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)(smem + base + img_off<DP>(rb, chunk) + half));
    return u.v;
}

__device__ __forceinline__ f32x4 mfma(const bf16x8 a, const bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
// three-term split product (A = Ah + Al, B = Bh + Bl; Al.Bl dropped): ~2^-16 relative
__device__ __forceinline__ f32x4 mfma3(const bf16x8 ah, const bf16x8 al, const bf16x8 bh, const bf16x8 bl, f32x4 c) {
    c = mfma(ah, bh, c);
    c = mfma(al, bh, c);
    c = mfma(ah, bl, c);
    return c;
}
// operand with NP parts (1 = exact bf16 data, 2 = hi + lo split)
template <int NP> struct Frag { bf16x8 p[NP]; };
template <int NA, int NB>
__device__ __forceinline__ f32x4 mfma_parts(const Frag<NA>& a, const Frag<NB>& b, f32x4 c) {
    c = mfma(a.p[0], b.p[0], c);
    if constexpr (NA == 2) c = mfma(a.p[1], b.p[0], c);
    if constexpr (NB == 2) c = mfma(a.p[0], b.p[1], c);
    return c;
}

// sum over the 16 lanes of a DPP row; the total lands in lane 15 of the row
__device__ __forceinline__ float row16_sum_to_lane15(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));
    return v;
}


typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <typename TIN> struct InTraits;
template <> struct InTraits<float> { static constexpr int EPL = 4, NP = 2; };
template <> struct InTraits<bf16_t> { static constexpr int EPL = 8, NP = 1; };
template <> struct InTraits<f16_t> { static constexpr int EPL = 8, NP = 2; };

// one 16-byte load of row `row` (clamped) at element column c*EPL; zero outside the tensor
// NT: non-temporal policy for tensors that are streamed exactly once (measured +5 % on the headline kernel)
template <typename TIN, bool NT = false>
__device__ __forceinline__ u32x4 load_piece(const TIN* base, int64_t sn, int row, int nrows, int c, int D) {
    constexpr int EPL = InTraits<TIN>::EPL;
    const bool ok = row < nrows && c * EPL < D;
    const int rr = row < nrows ? row : nrows - 1;
    const int cc = c * EPL < D ? c : 0;
    const u32x4* src = reinterpret_cast<const u32x4*>(base + (int64_t)rr * sn + cc * EPL);
    u32x4 v;
    if constexpr (NT) v = __builtin_nontemporal_load(src);
    else v = *src;
    if (!ok) v = u32x4{0, 0, 0, 0};
    return v;
}

// write one staged piece into the bf16 image(s) of a tile; part p lives at base + p*IMG
template <int DP, typename TIN, int SW = 0>
__device__ __forceinline__ void stage_piece(char* smem, int base, int row, int c, const u32x4 raw) {
    constexpr int IMG = img_bytes<DP, SW>();
    if constexpr (sizeof(TIN) == 4) {
        bf16x4 hi, lo;
        split4(__builtin_bit_cast(f32x4, raw), hi, lo);
        const int off = img_off<DP, SW>(row, c >> 1) + ((c & 1) << 3);
        *reinterpret_cast<bf16x4*>(smem + base + off) = hi;
        *reinterpret_cast<bf16x4*>(smem + base + IMG + off) = lo;
    } else if constexpr (InTraits<TIN>::NP == 1) {
        *reinterpret_cast<u32x4*>(smem + base + img_off<DP, SW>(row, c)) = raw;
    } else {
        typedef _Float16 h8 __attribute__((ext_vector_type(8)));
        const h8 hv = __builtin_bit_cast(h8, raw);
        f32x4 x0, x1;
#pragma unroll
        for (int i = 0; i < 4; ++i) { x0[i] = (float)hv[i]; x1[i] = (float)hv[4 + i]; }
        bf16x4 h0, l0, h1, l1;
        split4(x0, h0, l0);
        split4(x1, h1, l1);
        const int off = img_off<DP, SW>(row, c);
        *reinterpret_cast<bf16x8*>(smem + base + off) = cat4(h0, h1);
        *reinterpret_cast<bf16x8*>(smem + base + IMG + off) = cat4(l0, l1);
    }
}

__device__ __forceinline__ void store4_any(void* base, int dtype, int64_t idx, const f32x4 v) {
    // outputs are written once and never re-read by the kernel: non-temporal stores
    if (dtype == FASTMAX_F32) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(reinterpret_cast<float*>(base) + idx));
    else if (dtype == FASTMAX_BF16) __builtin_nontemporal_store(to_bf16x4(v), reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(base) + idx));
    else {
        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
        h4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = (_Float16)v[i];
        __builtin_nontemporal_store(o, reinterpret_cast<h4*>(reinterpret_cast<_Float16*>(base) + idx));
    }
}


// Stage a wave's 16 x DP fp32 accumulator tile (lane = row r, acc[dt][reg] = column 16dt + 4q4 + reg)
// through a private LDS area and write it out as whole rows in `dtype`.
template <int DP>
__device__ __forceinline__ void store_tile16(char* ost, const f32x4 (&acc)[DP / 16], float scale, int lane, void* out,
                                             int dtype, int64_t row0_elem, int first_row, int nrows, int D) {
    constexpr int C16 = DP / 4, DT = DP / 16;
    const int r = lane & 15, q4 = lane >> 4;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
        const int c16 = 4 * dt + q4;
        *reinterpret_cast<f32x4*>(ost + r * (DP * 4) + (((c16 ^ r) & (C16 - 1)) << 4)) = acc[dt] * scale;
    }
#pragma unroll
    for (int u = 0; u < (16 * C16) / 64; ++u) {
        const int idx = u * 64 + lane, rl = idx / C16, c16 = idx % C16;
        const f32x4 val = *reinterpret_cast<const f32x4*>(ost + rl * (DP * 4) + (((c16 ^ rl) & (C16 - 1)) << 4));
        if (first_row + rl < nrows && 4 * c16 < D)
            store4_any(out, dtype, row0_elem + (int64_t)rl * D + 4 * c16, val);
    }
}

// 16 bytes of input -> EPL floats
template <typename TIN> __device__ __forceinline__ void piece_to_float(const u32x4 raw, float (&x)[InTraits<TIN>::EPL]) {
    if constexpr (sizeof(TIN) == 4) {
        const f32x4 f = __builtin_bit_cast(f32x4, raw);
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i] = f[i];
    } else if constexpr (InTraits<TIN>::NP == 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            x[2 * i] = __uint_as_float(raw[i] << 16);
            x[2 * i + 1] = __uint_as_float(raw[i] & 0xffff0000u);
        }
    } else {
        typedef _Float16 h8 __attribute__((ext_vector_type(8)));
        const h8 hv = __builtin_bit_cast(h8, raw);
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = (float)hv[i];
    }
}

// EPL floats -> bf16 part images (NPI parts) at (row, piece column c)
template <int DP, int EPL, int NPI>
__device__ __forceinline__ void stage_floats(char* smem, int base, int row, int c, const float (&x)[EPL]) {
    constexpr int IMG = 64 * DP * 2;
#pragma unroll
    for (int hseg = 0; hseg < EPL / 4; ++hseg) {
        const f32x4 v = {x[4 * hseg], x[4 * hseg + 1], x[4 * hseg + 2], x[4 * hseg + 3]};
        const int e0 = c * EPL + 4 * hseg;                               // element column
        const int off = img_off<DP>(row, e0 >> 3) + ((e0 & 7) << 1);
        if constexpr (NPI == 2) {
            bf16x4 hi, lo;
            split4(v, hi, lo);
            *reinterpret_cast<bf16x4*>(smem + base + off) = hi;
            *reinterpret_cast<bf16x4*>(smem + base + IMG + off) = lo;
        } else {
            *reinterpret_cast<bf16x4*>(smem + base + off) = to_bf16x4(v);
        }
    }
}

// sum over the lanes that hold one staged row (COLS consecutive lanes, COLS in {8,16,32}); result valid in the
// LAST lane of the group
template <int COLS> __device__ __forceinline__ float rowgroup_sum(float v) {
    if constexpr (COLS == 32) {
        v = row16_sum_to_lane15(v);
        // lane 15 / 31 of each 32-lane half hold the two halves: fold lane 15 into lane 31 (row_bcast15 semantics via shuffle)
        const float lo = __shfl_up(v, 16, 64);
        return v + lo;
    } else if constexpr (COLS == 16) {
        return row16_sum_to_lane15(v);
    } else {
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true));
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true));
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));
        return v;
    }
}
// broadcast of the group total back to every lane of the group (for the fused mean subtraction)
template <int COLS> __device__ __forceinline__ float rowgroup_allsum(float v) {
#pragma unroll
    for (int off = 1; off < COLS; off <<= 1) v += __shfl_xor(v, off, 64);
    return v;
}




// sum over the COLS (8 or 16) consecutive lanes that hold one staged row, delivered to EVERY lane of the group, in DPP
// instructions only (quad permutes for the xor-1 / xor-2 steps, row_half_mirror and row_mirror for the 4- and 8-lane steps):
// 3 / 4 v_add_f32 instead of a ds_bpermute + add per step
template <int COLS> __device__ __forceinline__ float rowgroup_allsum_dpp(float v) {
    static_assert(COLS == 8 || COLS == 16, "one or half a DPP row");
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));   // row_half_mirror
    if constexpr (COLS == 16)
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true)); // row_mirror
    return v;
}

// sum over the COLS consecutive lanes that hold one staged row, in every lane: DPP adds where a row is 8 or 16 lanes
template <int COLS> __device__ __forceinline__ float rowsum_all(float v) {
    if constexpr (COLS == 8 || COLS == 16) return rowgroup_allsum_dpp<COLS>(v);
    else return rowgroup_allsum<COLS>(v);
}

// linearmax prologue on one staged 16-byte piece: (x - mean_D x) * scale as one fma per element (fastmax_hack.py:38-43); `sc_c`
// is the head's scale, zero on a padded head column; a row past N was loaded as zeros (mean 0 -> 0)
template <int COLS, int EPL> __device__ __forceinline__ void normalize_piece(float (&x)[EPL], float sc_c, float invD) {
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) s += x[e];
    const float nm = -rowsum_all<COLS>(s) * invD * sc_c;
#pragma unroll
    for (int e = 0; e < EPL; ++e) x[e] = fmaf(x[e], sc_c, nm);
}

// Same, but staged in the OUTPUT dtype through a wave-private area, so no workgroup barrier is needed: the 16 image
// rows a wave alone reads (its query rows) are free once their fragments sit in registers.  area0 / area1 are the
// two 16-row blocks (2*DP bytes per row) of a two-part image; a 4-byte result row is split across them, a 2-byte
// result row fits area0.  TOUT_BYTES = element size of the result.
template <int DP, int TOUT_BYTES>
__device__ __forceinline__ void store_tile16_private(char* area0, char* area1, const f32x4 (&acc)[DP / 16], float scale,
                                                     int lane, void* out, int dtype, int64_t row0_elem, int first_row,
                                                     int nrows, int D) {
    constexpr int DT = DP / 16, NCH = DP / 8;                 // 16-byte chunks per area row
    const int r = lane & 15, q4 = lane >> 4;
    if constexpr (TOUT_BYTES == 4) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const int c16 = 4 * dt + q4, cc = c16 % NCH;
            char* area = (c16 / NCH) ? area1 : area0;
            *reinterpret_cast<f32x4*>(area + r * (2 * DP) + (((cc ^ r) & (NCH - 1)) << 4)) = acc[dt] * scale;
        }
#pragma unroll
        for (int u = 0; u < (16 * (DP / 4)) / 64; ++u) {
            const int idx = u * 64 + lane, rl = idx / (DP / 4), c16 = idx % (DP / 4), cc = c16 % NCH;
            const char* area = (c16 / NCH) ? area1 : area0;
            const f32x4 val = *reinterpret_cast<const f32x4*>(area + rl * (2 * DP) + (((cc ^ rl) & (NCH - 1)) << 4));
            if (first_row + rl < nrows && 4 * c16 < D)
                __builtin_nontemporal_store(val, reinterpret_cast<f32x4*>(reinterpret_cast<float*>(out) + row0_elem + (int64_t)rl * D + 4 * c16));
        }
    } else {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const int c8 = 4 * dt + q4;                        // 8-byte unit inside the row
            char* dst = area0 + r * (2 * DP) + ((((c8 >> 1) ^ r) & (NCH - 1)) << 4) + ((c8 & 1) << 3);
            const f32x4 v = acc[dt] * scale;
            if (dtype == FASTMAX_BF16) *reinterpret_cast<bf16x4*>(dst) = to_bf16x4(v);
            else {
                typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                h4 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = (_Float16)v[i];
                *reinterpret_cast<h4*>(dst) = o;
            }
        }
#pragma unroll
        for (int u = 0; u < (16 * NCH + 63) / 64; ++u) {
            const int idx = u * 64 + lane, rl = idx / NCH, cc = idx % NCH;
            if (idx < 16 * NCH) {
                const u32x4 val = *reinterpret_cast<const u32x4*>(area0 + rl * (2 * DP) + (((cc ^ rl) & (NCH - 1)) << 4));
                if (first_row + rl < nrows && 8 * cc < D)
                    __builtin_nontemporal_store(val, reinterpret_cast<u32x4*>(reinterpret_cast<uint16_t*>(out) + row0_elem + (int64_t)rl * D + 8 * cc));
            }
        }
    }
}


// Walks 64-row tiles of one (b,h) slab of a (N, D) tensor with the staging map (row = srow + ps*RPP, piece scol):
// interior tiles are plain pointer-increment loads (no clamps, no compares); only a tile that runs past the
// tensor or a padded head size goes through load_piece.
template <typename TIN, int NPASS, int RPP, bool NT = false>
struct TileLoader {
    const TIN* base;
    int64_t sn;
    int nrows, D, srow, scol;
    bool full_cols;
    __device__ __forceinline__ TileLoader(const TIN* b, int64_t sn_, int nrows_, int D_, int DP, int srow_, int scol_)
        : base(b), sn(sn_), nrows(nrows_), D(D_), srow(srow_), scol(scol_), full_cols(D_ == DP) {}
    __device__ __forceinline__ void load(int tile, u32x4 (&r)[NPASS]) const {
        constexpr int EPL = InTraits<TIN>::EPL;
        const int row0 = tile * 64;
        if (full_cols && row0 + 64 <= nrows) {                       // block-uniform
            const TIN* p = base + (int64_t)(row0 + srow) * sn + scol * EPL;
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps) {
                const u32x4* src = reinterpret_cast<const u32x4*>(p + (int64_t)ps * RPP * sn);
                if constexpr (NT) r[ps] = __builtin_nontemporal_load(src);
                else r[ps] = *src;
            }
        } else {
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps) r[ps] = load_piece<TIN, NT>(base, sn, row0 + srow + ps * RPP, nrows, scol, D);
        }
    }
};


// 64-row tiles of one (b,h) slab of a (N, D) tensor through a buffer descriptor (staging map: row = srow + ps*RPP,
// 16-byte piece scol).  The record count is the slab's own byte range, so rows past the tensor read as zero in
// hardware (the range check covers the VGPR offset, which is why the tile offset is added there and not passed as the
// scalar offset); lanes of a padded head column carry an offset that is always out of range.  A request costs NPASS
// loads + NPASS integer adds, no compares.  The caller guarantees the slab spans less than 2 GiB (quad32_span_ok).
// AUX: cache-policy bits of the loads (2 = non-temporal: tensors that are streamed exactly once)
template <typename TIN, int NPASS, int RPP, int AUX = 0>
struct BufTileLoader {
    __amdgpu_buffer_rsrc_t rs;
    int voff[NPASS];
    int tile_bytes;
    __device__ __forceinline__ BufTileLoader(const TIN* base, int64_t sn, int nrows, int D, int srow, int scol) {
        constexpr int EPL = InTraits<TIN>::EPL;
        const int row_bytes = (int)sn * (int)sizeof(TIN);
        const int nrec = nrows > 0 ? (nrows - 1) * row_bytes + D * (int)sizeof(TIN) : 0;
        rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<TIN*>(base), 0, nrec, 0x00020000);
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps)
            voff[ps] = scol * EPL < D ? (srow + ps * RPP) * row_bytes + scol * 16 : (int)0x80000000;
        tile_bytes = 64 * row_bytes;
    }
    __device__ __forceinline__ void load(int tile, u32x4 (&r)[NPASS]) const {
        const int t = tile * tile_bytes;
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) r[ps] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff[ps] + t, 0, AUX);
    }
};
// the scan kernels' streamed-once tile loader: buffer descriptors where every slab fits their 31-bit offsets (BUF, decided on the
// host), else 64-bit pointer arithmetic
template <bool BUF, typename TIN, int NPASS, int RPP> struct ScanLoader;
// The buffer form for kernels that stream four or five tensors with every register taken (the linear-time backward at D = 128):
// nothing per (loader, pass) lives in a vector register across the chunk loop -- the offset of a piece is re-formed from the
// thread's row / column (shared by all loaders) with one 24-bit multiply-add and one add per load; the row stride, the tile
// stride and the descriptor are scalars.  BufTileLoader keeps NPASS offsets per loader: at four loaders the D = 128 dK/dV
// kernel spilled them, and a scratch reload in front of the prefetch loads is a full memory round trip per chunk.
template <typename TIN, int NPASS, int RPP, int AUX> struct BufTileLoaderLean {
    __amdgpu_buffer_rsrc_t rs;
    int row_bytes, base_off, srow;
    __device__ __forceinline__ BufTileLoaderLean(const TIN* base, int64_t sn, int nrows, int D, int srow_, int scol) : srow(srow_) {
        constexpr int EPL = InTraits<TIN>::EPL;
        row_bytes = __builtin_amdgcn_readfirstlane((int)sn * (int)sizeof(TIN));
        const int nrec = nrows > 0 ? (nrows - 1) * row_bytes + D * (int)sizeof(TIN) : 0;
        rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<TIN*>(base), 0, nrec, 0x00020000);
        base_off = scol * EPL < D ? scol * 16 : (int)0x80000000;       // a padded head column is always out of range
    }
    __device__ __forceinline__ void load(int tile, u32x4 (&r)[NPASS]) const {
        const int t = tile * 64 * row_bytes;
        // opaque copy of the row index: without it the compiler hoists the (loop-invariant) products out of the chunk loop,
        // runs out of registers, spills them and reloads each one with s_waitcnt vmcnt(0) between the prefetch loads
        int sr = srow;
        asm volatile("" : "+v"(sr));
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps)
            r[ps] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)__umul24((unsigned)(sr + ps * RPP), (unsigned)row_bytes) + base_off + t, 0, AUX);
    }
};
template <typename TIN, int NPASS, int RPP> struct ScanLoader<true, TIN, NPASS, RPP> : BufTileLoaderLean<TIN, NPASS, RPP, 2> {
    __device__ __forceinline__ ScanLoader(const TIN* b, int64_t sn, int nrows, int D, int, int srow, int scol)
        : BufTileLoaderLean<TIN, NPASS, RPP, 2>(b, sn, nrows, D, srow, scol) {}
};
// tile kernels: the offsets kept in registers where registers are plentiful (single-part operands), re-formed where they are not
template <typename TIN, int NPASS, int RPP>
using TileKernelLoader = std::conditional_t<InTraits<TIN>::NP == 2, BufTileLoaderLean<TIN, NPASS, RPP, 0>, BufTileLoader<TIN, NPASS, RPP>>;
template <typename TIN, int NPASS, int RPP> struct ScanLoader<false, TIN, NPASS, RPP> : TileLoader<TIN, NPASS, RPP, true> {
    __device__ __forceinline__ ScanLoader(const TIN* b, int64_t sn, int nrows, int D, int DP, int srow, int scol)
        : TileLoader<TIN, NPASS, RPP, true>(b, sn, nrows, D, DP, srow, scol) {}
};

// 16-byte pieces of single rows (a wave's query rows straight into B fragments): same two forms.  The pointer form costs a
// 64-bit address per piece; eight of them, hoisted out of the chunk loop and spilled, serialised the fp32 D = 128 forward's
// Q loads (scratch reload + s_waitcnt before every load).
template <bool BUF, typename TIN> struct RowPieceLoader;
template <typename TIN> struct RowPieceLoader<true, TIN> {
    __amdgpu_buffer_rsrc_t rs;
    int row_bytes, D;
    __device__ __forceinline__ RowPieceLoader(const TIN* base, int64_t sn, int nrows, int D_) : D(D_) {
        row_bytes = __builtin_amdgcn_readfirstlane((int)sn * (int)sizeof(TIN));
        const int nrec = nrows > 0 ? (nrows - 1) * row_bytes + D * (int)sizeof(TIN) : 0;
        rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<TIN*>(base), 0, nrec, 0x00020000);
    }
    // rows past the tensor and pieces past D read as zeros (range check of the descriptor)
    __device__ __forceinline__ u32x4 load(int row, int piece) const {
        constexpr int EPL = InTraits<TIN>::EPL;
        const int off = piece * EPL < D ? (int)__umul24((unsigned)row, (unsigned)row_bytes) + piece * 16 : (int)0x80000000;
        return __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 2);
    }
};
template <typename TIN> struct RowPieceLoader<false, TIN> {
    const TIN* base;
    int64_t sn;
    int nrows, D;
    __device__ __forceinline__ RowPieceLoader(const TIN* b, int64_t sn_, int nrows_, int D_) : base(b), sn(sn_), nrows(nrows_), D(D_) {}
    __device__ __forceinline__ u32x4 load(int row, int piece) const { return load_piece<TIN, true>(base, sn, row, nrows, piece, D); }
};

// host side: the byte range of one (b,h) slab fits the 31-bit offsets above
inline bool quad32_span_ok(int64_t sn, int nrows, int D, int elem_bytes) {
    return sn >= 0 && ((int64_t)(nrows > 0 ? nrows - 1 : 0) * sn + D) * elem_bytes < (int64_t)0x40000000;
}

}  // namespace fastmax
