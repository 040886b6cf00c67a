// Decode-time state cache for first-order fastmax (opt-in):
// instead of re-running unmasked attention over a zero-padded KV cache for every generated token
// (lit_gpt/model.py:427-430, 464-466; generate/base.py:85-92) the per-head carried state
//     S2 = sum_j k_j v_j^T (D x D),  S1 = sum_j v_j,  ksum = sum_j k_j,  count
// is kept in HBM: a prefill pass produces it from the prompt (the segment-state kernel of
// fastmax_mfma_split.hip), and each new token costs O(D^2) -- update the state, read out
//     o = (S1 + a S2^T q) / (count + a q.ksum).
// This equals masked p=1 fastmax at the last position of the extended sequence.  It is NOT the
// reference's decode arithmetic (unmasked over the padded cache with constant N_q, quirk Q4), hence opt-in.
#include "fastmax_mfma_common.h"

namespace fastmax {

// one workgroup per (b,h); record = [S2 (DP x DP, [m][d]) | S1 (DP) | ksum (DP)], fp32
template <typename T, int DP>
__global__ __launch_bounds__(256) void p1_decode_step_kernel(const void* q, const void* k, const void* v, Strides3 qs, Strides3 ks,
                                                             Strides3 vs, float* state, void* o, int out_dtype, int H, int D,
                                                             float a, float count_after) {
    constexpr int RG = 256 / DP, RPT = DP / RG;                 // row groups, rows per thread
    __shared__ float qv[DP], kv[DP], vv[DP], part[RG][DP], gpart[4];
    const int tid = threadIdx.x, bh = blockIdx.x, b = bh / H, h = bh % H;
    float* rec = state + (int64_t)bh * (DP * DP + 2 * DP);
    if (tid < DP) {
        const bool ok = tid < D;
        qv[tid] = ok ? to_float(row_ptr<T>(q, qs.sb, qs.sh, qs.sn, b, h, 0)[tid]) : 0.f;
        kv[tid] = ok ? to_float(row_ptr<T>(k, ks.sb, ks.sh, ks.sn, b, h, 0)[tid]) : 0.f;
        vv[tid] = ok ? to_float(row_ptr<T>(v, vs.sb, vs.sh, vs.sn, b, h, 0)[tid]) : 0.f;
    }
    __syncthreads();
    const int d = tid % DP, rg = tid / DP;
    float f = 0.f;
#pragma unroll 4
    for (int i = 0; i < RPT; ++i) {
        const int m = rg * RPT + i;
        const float s = fmaf(kv[m], vv[d], rec[m * DP + d]);    // S2[m][d] += k_m v_d
        rec[m * DP + d] = s;
        f = fmaf(qv[m], s, f);
    }
    part[rg][d] = f;
    // g = count + a q.ksum (ksum updated first)
    float gp = 0.f;
    if (tid < DP) {
        const float ks_new = rec[DP * DP + DP + tid] + kv[tid];
        rec[DP * DP + DP + tid] = ks_new;
        gp = qv[tid] * ks_new;
    }
    gp = wave_sum(gp);
    if ((tid & 63) == 0) gpart[tid >> 6] = gp;
    __syncthreads();
    if (tid < DP) {
        const float s1 = rec[DP * DP + tid] + vv[tid];
        rec[DP * DP + tid] = s1;
        float fs = 0.f;
#pragma unroll
        for (int g = 0; g < RG; ++g) fs += part[g][tid];
        const float gval = count_after + a * (gpart[0] + gpart[1] + gpart[2] + gpart[3]);
        if (tid < D) {
            const float val = (s1 + a * fs) / gval;
            const int64_t idx = (int64_t)bh * D + tid;
            if (out_dtype == FASTMAX_F32) reinterpret_cast<float*>(o)[idx] = val;
            else if (out_dtype == FASTMAX_BF16) reinterpret_cast<uint16_t*>(o)[idx] = f32_to_bf16_bits(val);
            else reinterpret_cast<_Float16*>(o)[idx] = (_Float16)val;
        }
    }
}

template <typename T>
static int launch_decode_t(const void* q, const void* k, const void* v, Strides3 qs, Strides3 ks, Strides3 vs, float* state,
                           void* o, int out_dtype, int B, int H, int D, float a, float count_after, hipStream_t stream) {
    if (D <= 64)
        hipLaunchKernelGGL((p1_decode_step_kernel<T, 64>), dim3(B * H), dim3(256), 0, stream, q, k, v, qs, ks, vs, state, o,
                           out_dtype, H, D, a, count_after);
    else
        hipLaunchKernelGGL((p1_decode_step_kernel<T, 128>), dim3(B * H), dim3(256), 0, stream, q, k, v, qs, ks, vs, state, o,
                           out_dtype, H, D, a, count_after);
    return (int)hipGetLastError();
}

}  // namespace fastmax

using namespace fastmax;

extern "C" {

size_t fastmax_hip_decode_state_bytes(int B, int H, int D) {
    if (B <= 0 || H <= 0 || D <= 0 || D > 128) return 0;                 // the state cache carries a D x D state: D <= 128
    const size_t dp = D <= 64 ? 64 : 128;
    return sizeof(float) * (size_t)B * H * (dp * dp + 2 * dp);
}

int fastmax_hip_p1_prefill_state(const fastmax_problem* prob, const void* k, const int64_t* k_strides, const void* v,
                                 const int64_t* v_strides, float* state, void* stream) {
    if (!prob || !k || !v || !state || !k_strides || !v_strides) return FASTMAX_E_NULL;
    if (prob->p != 1 || !prob->causal) return FASTMAX_E_BAD_P;
    if (!mfma_gen_supported(*prob, false)) return FASTMAX_E_BAD_SHAPE;
    const int dp = prob->D <= 64 ? 64 : 128;
    const int nchunks = (prob->Nq + 63) / 64;
    FwdArgs a{*prob, nullptr, k, v, Strides3{0, 0, 0}, Strides3{k_strides[0], k_strides[1], k_strides[2]},
              Strides3{v_strides[0], v_strides[1], v_strides[2]}, nullptr, nullptr, state, fastmax_hip_decode_state_bytes(prob->B, prob->H, prob->D),
              reinterpret_cast<hipStream_t>(stream)};
    // one "segment" that covers the whole prompt: nseg = 2 -> the kernel runs segment 0 only
    return launch_split_states(a, SplitPlan{2, nchunks}, dp, nullptr);
}

int fastmax_hip_p1_decode_step(const void* q, const int64_t* q_strides, const void* k, const int64_t* k_strides, const void* v,
                               const int64_t* v_strides, float* state, void* o, int B, int H, int D, int in_dtype,
                               int out_dtype, float a, int64_t count_after, void* stream) {
    if (!q || !k || !v || !state || !o || !q_strides || !k_strides || !v_strides) return FASTMAX_E_NULL;
    if (B <= 0 || H <= 0 || D <= 0 || D > 128 || count_after <= 0) return FASTMAX_E_BAD_SHAPE;
    const Strides3 qs{q_strides[0], q_strides[1], q_strides[2]}, ks{k_strides[0], k_strides[1], k_strides[2]},
        vs{v_strides[0], v_strides[1], v_strides[2]};
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    switch (in_dtype) {
        case FASTMAX_F32: return launch_decode_t<float>(q, k, v, qs, ks, vs, state, o, out_dtype, B, H, D, a, (float)count_after, st);
        case FASTMAX_BF16: return launch_decode_t<bf16_t>(q, k, v, qs, ks, vs, state, o, out_dtype, B, H, D, a, (float)count_after, st);
        case FASTMAX_F16: return launch_decode_t<f16_t>(q, k, v, qs, ks, vs, state, o, out_dtype, B, H, D, a, (float)count_after, st);
    }
    return FASTMAX_E_BAD_DTYPE;
}

}  // extern "C"
