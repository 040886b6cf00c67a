// Loss path of the fine-tune step: row-wise cross entropy over the vocabulary, the op behind the
// reference's chunked_cross_entropy (lit_gpt/utils.py:228-272, torch.nn.functional.cross_entropy with reduction="none" and
// ignore_index).  One workgroup per row, one pass over the logits each way:
//   forward : lse_i = log sum_v exp(z_iv),  loss_i = lse_i - z_i,t_i   (0 where t_i == ignore_index)
//   backward: dz_iv = (exp(z_iv - lse_i) - [v == t_i]) * gscale_i      (0 rows where ignored), written in place or to dz
// Logits are float32 / bf16 / fp16, read with 16-byte loads; the running (max, sum) pair is combined across the lanes
// and waves of the block.  HBM-bound: one read (forward), one read + one write (backward) of the (M, V) logits.
#include "fastmax_common.h"

namespace fastmax {

typedef unsigned int cu32x4 __attribute__((ext_vector_type(4)));

template <typename T> __device__ __forceinline__ void ce_load(const T* row, int64_t v0, int V, float (&x)[16 / sizeof(T)], bool vec) {
    constexpr int E = 16 / sizeof(T);
    if (vec && v0 + E <= V) {
        cu32x4 raw = __builtin_nontemporal_load(reinterpret_cast<const cu32x4*>(row + v0));
        const T* pv = reinterpret_cast<const T*>(&raw);
#pragma unroll
        for (int e = 0; e < E; ++e) x[e] = to_float(pv[e]);
    } else {
#pragma unroll
        for (int e = 0; e < E; ++e) x[e] = (v0 + e < V) ? to_float(row[v0 + e]) : -INFINITY;
    }
}

__device__ __forceinline__ void ms_combine(float& m, float& s, float m2, float s2) {
    const float mm = fmaxf(m, m2);
    // exp(-inf - -inf) would be NaN: an empty partial has s == 0 and contributes nothing
    const float a = s > 0.f ? s * __expf(m - mm) : 0.f, b = s2 > 0.f ? s2 * __expf(m2 - mm) : 0.f;
    m = mm;
    s = a + b;
}

template <typename T>
__global__ __launch_bounds__(256) void ce_fwd_kernel(const T* logits, int64_t ld, const int64_t* targets, float* loss, float* lse,
                                                     int V, int64_t ignore_index, int vec) {
    constexpr int E = 16 / sizeof(T);
    __shared__ float sm[4], ss[4];
    const int64_t i = blockIdx.x;
    const T* row = logits + i * ld;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float m = -INFINITY, s = 0.f;
    for (int64_t v0 = (int64_t)tid * E; v0 < V; v0 += 256 * E) {
        float x[E];
        ce_load<T>(row, v0, V, x, vec);
        float cm = x[0];
#pragma unroll
        for (int e = 1; e < E; ++e) cm = fmaxf(cm, x[e]);
        float cs = 0.f;
#pragma unroll
        for (int e = 0; e < E; ++e) cs += __expf(x[e] - cm);
        ms_combine(m, s, cm, cs);
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float m2 = __shfl_xor(m, off, 64), s2 = __shfl_xor(s, off, 64);
        ms_combine(m, s, m2, s2);
    }
    if (lane == 0) { sm[wave] = m; ss[wave] = s; }
    __syncthreads();
    if (tid == 0) {
        float mm = sm[0], st = ss[0];
        for (int w = 1; w < 4; ++w) ms_combine(mm, st, sm[w], ss[w]);
        const float l = mm + __logf(st);
        const int64_t t = targets[i];
        lse[i] = l;
        loss[i] = (t == ignore_index || t < 0 || t >= V) ? 0.f : l - to_float(row[t]);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void ce_bwd_kernel(const T* logits, int64_t ld, const int64_t* targets, const float* lse,
                                                     const float* gscale, float gconst, T* dz, int64_t ldz, int V,
                                                     int64_t ignore_index, int vec) {
    constexpr int E = 16 / sizeof(T);
    const int64_t i = blockIdx.x;
    const T* row = logits + i * ld;
    T* out = dz + i * ldz;
    const int64_t t = targets[i];
    const bool ignored = (t == ignore_index || t < 0 || t >= V);
    const float g = ignored ? 0.f : (gscale ? gscale[i] : 1.0f) * gconst;
    const float l = lse[i];
    for (int64_t v0 = (int64_t)threadIdx.x * E; v0 < V; v0 += 256 * E) {
        float x[E];
        ce_load<T>(row, v0, V, x, vec);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const float p = (v0 + e < V) ? __expf(x[e] - l) : 0.f;
            x[e] = (p - ((v0 + e == t) ? 1.0f : 0.f)) * g;
        }
        if (vec && v0 + E <= V) {
            cu32x4 raw;
            T* pv = reinterpret_cast<T*>(&raw);
#pragma unroll
            for (int e = 0; e < E; ++e) pv[e] = from_float<T>(x[e]);
            *reinterpret_cast<cu32x4*>(out + v0) = raw;
        } else {
#pragma unroll
            for (int e = 0; e < E; ++e)
                if (v0 + e < V) out[v0 + e] = from_float<T>(x[e]);
        }
    }
}

template <typename T>
static int ce_launch(const void* logits, int64_t ld, const int64_t* targets, float* loss, float* lse, const float* gscale, float gconst,
                     void* dz, int64_t ldz, int64_t M, int V, int64_t ignore_index, bool bwd, hipStream_t stream) {
    const int es = (int)sizeof(T);
    const int vec = !(reinterpret_cast<uintptr_t>(logits) & 15) && ((ld * es) % 16 == 0) &&
                    (!bwd || (!(reinterpret_cast<uintptr_t>(dz) & 15) && ((ldz * es) % 16 == 0)));
    if (M > 0x7fffffff) return FASTMAX_E_BAD_SHAPE;
    if (!bwd)
        hipLaunchKernelGGL((ce_fwd_kernel<T>), dim3((unsigned)M), dim3(256), 0, stream, reinterpret_cast<const T*>(logits), ld, targets,
                           loss, lse, V, ignore_index, vec);
    else
        hipLaunchKernelGGL((ce_bwd_kernel<T>), dim3((unsigned)M), dim3(256), 0, stream, reinterpret_cast<const T*>(logits), ld, targets,
                           lse, gscale, gconst, reinterpret_cast<T*>(dz), ldz, V, ignore_index, vec);
    return (int)hipGetLastError();
}

}  // namespace fastmax

using namespace fastmax;

extern "C" {

int fastmax_hip_cross_entropy_forward(const void* logits, int64_t ld, const int64_t* targets, float* loss, float* lse, int64_t M,
                                      int V, int64_t ignore_index, int dtype, void* stream) {
    if (!logits || !targets || !loss || !lse) return FASTMAX_E_NULL;
    if (M <= 0 || V <= 0 || ld < V) return FASTMAX_E_BAD_SHAPE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    switch (dtype) {
        case FASTMAX_F32: return ce_launch<float>(logits, ld, targets, loss, lse, nullptr, 1.f, nullptr, 0, M, V, ignore_index, false, s);
        case FASTMAX_BF16: return ce_launch<bf16_t>(logits, ld, targets, loss, lse, nullptr, 1.f, nullptr, 0, M, V, ignore_index, false, s);
        case FASTMAX_F16: return ce_launch<f16_t>(logits, ld, targets, loss, lse, nullptr, 1.f, nullptr, 0, M, V, ignore_index, false, s);
    }
    return FASTMAX_E_BAD_DTYPE;
}

int fastmax_hip_cross_entropy_backward(const void* logits, int64_t ld, const int64_t* targets, const float* lse,
                                       const float* grad_loss, float grad_scale, void* grad_logits, int64_t ldg, int64_t M, int V,
                                       int64_t ignore_index, int dtype, void* stream) {
    if (!logits || !targets || !lse || !grad_logits) return FASTMAX_E_NULL;
    if (M <= 0 || V <= 0 || ld < V || ldg < V) return FASTMAX_E_BAD_SHAPE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    switch (dtype) {
        case FASTMAX_F32: return ce_launch<float>(logits, ld, targets, nullptr, const_cast<float*>(lse), grad_loss, grad_scale, grad_logits, ldg, M, V, ignore_index, true, s);
        case FASTMAX_BF16: return ce_launch<bf16_t>(logits, ld, targets, nullptr, const_cast<float*>(lse), grad_loss, grad_scale, grad_logits, ldg, M, V, ignore_index, true, s);
        case FASTMAX_F16: return ce_launch<f16_t>(logits, ld, targets, nullptr, const_cast<float*>(lse), grad_loss, grad_scale, grad_logits, ldg, M, V, ignore_index, true, s);
    }
    return FASTMAX_E_BAD_DTYPE;
}

}  // extern "C"
