// The operator's neighbours in CausalSelfAttention.forward (lit_gpt/model.py:397-425), SURVEY.md 8(f) row 1, as ONE pass:
//   qkv (B, T, G, q_per_kv + 2, hs)  --de-interleave, RoPE on the first rope_n elements of q and k, GQA expand-->
//   q (B, H, T, hs), k, v (B, H or G, T, hs)
// instead of view / permute / split / expand / reshape copies + apply_rope's five elementwise launches + two cats, and the
// mirror image for the gradients (sum of dK, dV over the query heads of a group, inverse rotation, re-interleave).
// RoPE (model.py:702-708): out = x cos + rot(x) sin with rot(x) = cat(-x[half:], x[:half]); cos, sin: (T, rope_n) float32.
// One lane owns a 16-byte piece of the first half of the rotated range and its partner piece in the second half, or a
// 16-byte piece of the pass-through tail, of one head row at a time; a wave walks the head rows of one token.  rope_n / 2 and hs - rope_n must be multiples of the 16-byte element count.
#include "fastmax_common.h"

namespace fastmax {

typedef unsigned int ru32x4 __attribute__((ext_vector_type(4)));

struct RopeParams {
    const void* qkv;          // forward input / backward output (B,T,G,qpk+2,hs)
    void *q, *k, *v;          // forward outputs / backward inputs (gradients)
    const float *cos, *sin;
    int B, T, G, qpk, hs, rope_n, expand_kv;
    int tables16;             // the caller's rope cache was in the tensors' own 16-bit dtype ("bf16-true" precision): the two
                              // products are rounded to that dtype before the sum, as the tensor ops of model.py:708 then do
};

template <typename T, int E> __device__ __forceinline__ void ld_piece(const T* p, float (&x)[E]) {
    ru32x4 raw = *reinterpret_cast<const ru32x4*>(p);
    const T* pv = reinterpret_cast<const T*>(&raw);
#pragma unroll
    for (int e = 0; e < E; ++e) x[e] = to_float(pv[e]);
}
template <typename T, int E> __device__ __forceinline__ void st_piece(T* p, const float (&x)[E]) {
    ru32x4 raw;
    T* pv = reinterpret_cast<T*>(&raw);
#pragma unroll
    for (int e = 0; e < E; ++e) pv[e] = from_float<T>(x[e]);
    *reinterpret_cast<ru32x4*>(p) = raw;
}

// x cos + y sin with the reference's three float32 roundings (no fused multiply-add; hipcc contracts by default)
__device__ __forceinline__ float mul_add_unfused(float x, float c, float y, float s) {
#pragma clang fp contract(off)
    const float p0 = x * c;
    const float p1 = y * s;
    return p0 + p1;
}

// One wave = one token (b, t): its G * (qpk + 2) head rows x UPR units per row (a unit = a rotated piece pair or one tail
// piece) are walked 64 units at a time.  When UPR divides 64 a lane keeps the same unit for every row, so the four table
// pieces it needs (cos, sin at d and d + half: 4 x 16-byte-count floats) are loaded ONCE per token and lane -- fetched per row they
// are four times the bytes of the data itself (measured: 2 TB/s on the Llama-2-7B layout with per-row table loads).
// grid.x = ceil(B*T / 4), block = 256.
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void rope_qkv_kernel(RopeParams prm) {
    constexpr int E = 16 / sizeof(T);
    const int hs = prm.hs, half = prm.rope_n / 2, total = prm.qpk + 2;
    const int pair_units = half / E, tail_units = (hs - prm.rope_n) / E, upr = pair_units + tail_units;
    const int lane = threadIdx.x & 63;
    const int64_t tok = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tok >= (int64_t)prm.B * prm.T) return;
    const int t = (int)(tok % prm.T);
    const int b = (int)(tok / prm.T);
    const int H = prm.G * prm.qpk;
    const int nunits = prm.G * total * upr;
    const bool fixed_unit = (64 % upr) == 0;
    const float* cs = prm.cos + (int64_t)t * prm.rope_n;
    const float* sn = prm.sin + (int64_t)t * prm.rope_n;
    float c0[E], c1[E], s0[E], s1[E];
    auto load_tables = [&](int unit) {
        if (unit < pair_units) {
            const int d = unit * E;
#pragma unroll
            for (int e = 0; e < E; ++e) { c0[e] = cs[d + e]; c1[e] = cs[d + half + e]; s0[e] = sn[d + e]; s1[e] = sn[d + half + e]; }
        }
    };
    if (fixed_unit) load_tables(lane % upr);
    T* const qkv_tok = reinterpret_cast<T*>(const_cast<void*>(prm.qkv)) + tok * prm.G * total * hs;
    const int64_t hstride = (int64_t)prm.T * hs;

    for (int idx = lane; idx < nunits; idx += 64) {
        // row order: the 2 G key / value rows first, then the query rows -- in the backward pass a key / value row sums q_per_kv
        // gradient rows, and with all of them inside one 64-unit step the other steps carry light rows only
        const int unit = idx % upr, ord = idx / upr;
        const int slot = ord < 2 * prm.G ? prm.qpk + (ord & 1) : (ord - 2 * prm.G) % prm.qpk;
        const int g = ord < 2 * prm.G ? (ord >> 1) : (ord - 2 * prm.G) / prm.qpk;
        if (!fixed_unit) load_tables(unit);
        T* qkv_row = qkv_tok + (int64_t)(g * total + slot) * hs;
        const bool is_q = slot < prm.qpk, is_k = slot == prm.qpk;
        // destination(s) (forward) / source(s) (backward) in the (B, heads, T, hs) tensors
        T* base = reinterpret_cast<T*>(is_q ? prm.q : (is_k ? prm.k : prm.v));
        // expand_kv: 0 = k, v stay at their G heads, 1 = both repeated for the q_per_kv query heads of their group, 2 = only v
        const bool expand = !is_q && (prm.expand_kv == 1 || (prm.expand_kv == 2 && !is_k));
        const int heads = (is_q || expand) ? H : prm.G;
        const int h0 = is_q ? g * prm.qpk + slot : (expand ? g * prm.qpk : g);
        const int ncopy = expand ? prm.qpk : 1;
        T* hrow = base + (((int64_t)b * heads + h0) * prm.T + t) * hs;
        const bool rotate = (is_q || is_k) && unit < pair_units;     // v is never rotated

        if (unit >= pair_units) {                                    // pass-through tail piece
            const int d = prm.rope_n + (unit - pair_units) * E;
            float x[E];
            if constexpr (!BWD) {
                ld_piece<T, E>(qkv_row + d, x);
                for (int c = 0; c < ncopy; ++c) st_piece<T, E>(hrow + c * hstride + d, x);
            } else {
#pragma unroll
                for (int e = 0; e < E; ++e) x[e] = 0.f;
                for (int c = 0; c < ncopy; ++c) {
                    float y[E];
                    ld_piece<T, E>(hrow + c * hstride + d, y);
#pragma unroll
                    for (int e = 0; e < E; ++e) x[e] += y[e];
                }
                st_piece<T, E>(qkv_row + d, x);
            }
            continue;
        }
        const int d = unit * E;                                      // first-half piece; partner at d + half
        float lo[E], hi[E];
        if constexpr (!BWD) {
            ld_piece<T, E>(qkv_row + d, lo);
            ld_piece<T, E>(qkv_row + d + half, hi);
        } else {
#pragma unroll
            for (int e = 0; e < E; ++e) { lo[e] = 0.f; hi[e] = 0.f; }
            // the q_per_kv gradients of a key / value row: four rows' loads in flight at a time (a trip count known only at
            // run time would otherwise chain load -> add -> load)
            for (int c = 0; c < ncopy; c += 4) {
                ru32x4 ra[4], rb[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int cc = min(c + u, ncopy - 1);
                    ra[u] = *reinterpret_cast<const ru32x4*>(hrow + cc * hstride + d);
                    rb[u] = *reinterpret_cast<const ru32x4*>(hrow + cc * hstride + d + half);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (c + u < ncopy) {
                        const T* pa = reinterpret_cast<const T*>(&ra[u]);
                        const T* pb = reinterpret_cast<const T*>(&rb[u]);
#pragma unroll
                        for (int e = 0; e < E; ++e) { lo[e] += to_float(pa[e]); hi[e] += to_float(pb[e]); }
                    }
                }
            }
        }
        if (rotate) {
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const float a = lo[e], bb = hi[e];
                if constexpr (!BWD) {
                    // out[d] = x[d] cos[d] - x[d+half] sin[d];  out[d+half] = x[d+half] cos[d+half] + x[d] sin[d+half]
                    // with the reference's roundings (model.py:708: two float32 products, one float32 sum, one rounding to
                    // the tensor dtype) -- no fused multiply-add, so 16-bit results are bit-identical to the tensor ops
                    if (sizeof(T) == 2 && prm.tables16) {
                        lo[e] = to_float(from_float<T>(a * c0[e])) + to_float(from_float<T>(-bb * s0[e]));
                        hi[e] = to_float(from_float<T>(bb * c1[e])) + to_float(from_float<T>(a * s1[e]));
                    } else {
                        lo[e] = mul_add_unfused(a, c0[e], -bb, s0[e]);
                        hi[e] = mul_add_unfused(bb, c1[e], a, s1[e]);
                    }
                } else {
                    // transpose of the map above
                    lo[e] = a * c0[e] + bb * s1[e];
                    hi[e] = bb * c1[e] - a * s0[e];
                }
            }
        }
        if constexpr (!BWD) {
            for (int c = 0; c < ncopy; ++c) {
                st_piece<T, E>(hrow + c * hstride + d, lo);
                st_piece<T, E>(hrow + c * hstride + d + half, hi);
            }
        } else {
            st_piece<T, E>(qkv_row + d, lo);
            st_piece<T, E>(qkv_row + d + half, hi);
        }
    }
}

template <typename T>
static int launch_rope_t(const RopeParams& prm, bool bwd, hipStream_t stream) {
    constexpr int E = 16 / sizeof(T);
    const int64_t blocks = ((int64_t)prm.B * prm.T + 3) / 4;                  // one wave per token
    if (blocks > 0x7fffffff || (int64_t)prm.G * (prm.qpk + 2) * prm.hs > 0x7fffffff) return FASTMAX_E_BAD_SHAPE;
    if (bwd) hipLaunchKernelGGL((rope_qkv_kernel<T, true>), dim3((unsigned)blocks), dim3(256), 0, stream, prm);
    else hipLaunchKernelGGL((rope_qkv_kernel<T, false>), dim3((unsigned)blocks), dim3(256), 0, stream, prm);
    return (int)hipGetLastError();
}

int launch_rope_qkv(const RopeParams& prm, int dtype, bool bwd, hipStream_t stream) {
    switch (dtype) {
        case FASTMAX_F32: return launch_rope_t<float>(prm, bwd, stream);
        case FASTMAX_BF16: return launch_rope_t<bf16_t>(prm, bwd, stream);
        case FASTMAX_F16: return launch_rope_t<f16_t>(prm, bwd, stream);
    }
    return FASTMAX_E_BAD_DTYPE;
}

}  // namespace fastmax

using namespace fastmax;

static int rope_check(const void* qkv, const void* q, const void* k, const void* v, const float* c, const float* s, int B, int T,
                      int G, int qpk, int hs, int rope_n, int dtype) {
    if (!qkv || !q || !k || !v || !c || !s) return FASTMAX_E_NULL;
    if (B <= 0 || T <= 0 || G <= 0 || qpk <= 0 || hs <= 0 || rope_n < 0 || rope_n > hs || (rope_n & 1)) return FASTMAX_E_BAD_SHAPE;
    if (dtype < 0 || dtype > FASTMAX_F16) return FASTMAX_E_BAD_DTYPE;
    const int e = dtype == FASTMAX_F32 ? 4 : 8;
    if ((rope_n / 2) % e || (hs - rope_n) % e) return FASTMAX_E_BAD_SHAPE;
    if ((reinterpret_cast<uintptr_t>(qkv) | reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k) |
         reinterpret_cast<uintptr_t>(v)) & 15)
        return FASTMAX_E_ALIGNMENT;
    return 0;
}

extern "C" {

int fastmax_hip_rope_qkv_split(const void* qkv, const float* cos, const float* sin, void* q, void* k, void* v, int B, int T, int G,
                               int q_per_kv, int head_size, int rope_n_elem, int expand_kv, int dtype, void* stream) {
    const int rc = rope_check(qkv, q, k, v, cos, sin, B, T, G, q_per_kv, head_size, rope_n_elem, dtype);
    if (rc) return rc;
    RopeParams prm{qkv, q, k, v, cos, sin, B, T, G, q_per_kv, head_size, rope_n_elem, expand_kv & 3, (expand_kv >> 4) & 1};
    return launch_rope_qkv(prm, dtype, false, reinterpret_cast<hipStream_t>(stream));
}

int fastmax_hip_rope_qkv_split_backward(const void* grad_q, const void* grad_k, const void* grad_v, const float* cos,
                                        const float* sin, void* grad_qkv, int B, int T, int G, int q_per_kv, int head_size,
                                        int rope_n_elem, int expand_kv, int dtype, void* stream) {
    const int rc = rope_check(grad_qkv, grad_q, grad_k, grad_v, cos, sin, B, T, G, q_per_kv, head_size, rope_n_elem, dtype);
    if (rc) return rc;
    RopeParams prm{grad_qkv, const_cast<void*>(grad_q), const_cast<void*>(grad_k), const_cast<void*>(grad_v), cos, sin, B, T, G,
                   q_per_kv, head_size, rope_n_elem, expand_kv & 3, 0};
    return launch_rope_qkv(prm, dtype, true, reinterpret_cast<hipStream_t>(stream));
}

}  // extern "C"
