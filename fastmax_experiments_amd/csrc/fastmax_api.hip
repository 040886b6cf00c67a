// extern "C" entry points of libfastmax_hip.so (see include/fastmax_hip.h) + path selection.
#include "fastmax_common.h"

#include <climits>
#include <cstdlib>
#include <cstring>

using namespace fastmax;

namespace {
// bf16 problems take the all-MFMA kernel; FASTMAX_BF16_KERNEL=gen (tuning key "bf16_kernel" = 0) keeps the generic one
bool use_bf16_kernel(const fastmax_problem& p) {
    return mfma_bf16_supported(p) && tune_get(TUNE_BF16_KERNEL) != 0;
}
int validate(const fastmax_problem* p) {
    if (!p) return FASTMAX_E_NULL;
    if (p->p != 1 && p->p != 2) return FASTMAX_E_BAD_P;
    if (p->B <= 0 || p->H <= 0 || p->Nq <= 0 || p->Nk <= 0 || p->D <= 0 || p->D > FASTMAX_MAX_D)
        return FASTMAX_E_BAD_SHAPE;
    if (p->causal && p->Nq != p->Nk) return FASTMAX_E_BAD_SHAPE;
    if ((int64_t)p->B * p->H > 65535) return FASTMAX_E_BAD_SHAPE;       // (b,h) rides on gridDim.y in the tile kernels
    if (p->in_dtype < 0 || p->in_dtype > 2 || p->out_dtype < 0 || p->out_dtype > 2) return FASTMAX_E_BAD_DTYPE;
    return FASTMAX_OK;
}
Strides3 st(const int64_t* s) { return Strides3{s[0], s[1], s[2]}; }

int select(const fastmax_problem& p) {
    if (p.path == FASTMAX_PATH_QUADRATIC) return FASTMAX_PATH_QUADRATIC;
    if (p.path == FASTMAX_PATH_QUADRATIC_MFMA) return quad_mfma_supported(p) ? FASTMAX_PATH_QUADRATIC_MFMA : FASTMAX_E_BAD_SHAPE;
    const bool lin = (p.p == 1 && p.causal);
    // unmasked first order at sizes where a pass over K, V + one D x D product per query row beats the O(N_q N_k) tiles
    if (p.p == 1 && !p.causal && (p.path == FASTMAX_PATH_AUTO || p.path == FASTMAX_PATH_MFMA) && unmasked_lin_supported(p)) return FASTMAX_PATH_MFMA;
    // head sizes above 128 (pythia-1b, Gemma, stablelm-3b in lit_gpt/config.py): tile kernels only -- no D x D state is carried
    if (p.D > 128) {
        if (p.path == FASTMAX_PATH_RECURRENT || p.path == FASTMAX_PATH_MFMA) return FASTMAX_E_BAD_SHAPE;
        return quad_mfma_supported(p) ? FASTMAX_PATH_QUADRATIC_MFMA : FASTMAX_PATH_QUADRATIC;
    }
    if (p.path == FASTMAX_PATH_RECURRENT) return lin ? FASTMAX_PATH_RECURRENT : FASTMAX_E_BAD_SHAPE;
    const bool lin_mfma = lin && (mfma_p1_supported(p) || mfma_gen_supported(p, false) || mfma_d128_2p_supported(p));
    if (p.path == FASTMAX_PATH_MFMA) return lin_mfma ? FASTMAX_PATH_MFMA : FASTMAX_E_BAD_SHAPE;
    if (lin && lin_mfma) return FASTMAX_PATH_MFMA;
    // p = 1 masked shapes the linear-time matrix-core kernels do not cover (two-part operands at D > 64: their images and
    // the D x D state do not fit 160 KB of LDS): the matrix-core tiles beat the vector-ALU recurrence up to N ~ 20 k
    // (measured at D = 128: 2.5 vs 8.5 ms at N = 4096, 27 vs 34 ms at N = 16384)
    if (lin) return (quad_mfma_supported(p) && p.Nq <= 20000) ? FASTMAX_PATH_QUADRATIC_MFMA : FASTMAX_PATH_RECURRENT;
    return quad_mfma_supported(p) ? FASTMAX_PATH_QUADRATIC_MFMA : FASTMAX_PATH_QUADRATIC;
}
bool aligned16(const void* ptr, const int64_t* s, int dtype) {
    const int64_t es = dtype == FASTMAX_F32 ? 4 : 2;
    if (reinterpret_cast<uintptr_t>(ptr) & 15) return false;
    for (int i = 0; i < 3; ++i)
        if ((s[i] * es) & 15) return false;
    return true;
}
}  // namespace

namespace fastmax {
// "mfma_variant" numbers whose kernels leave out matrix instructions or memory passes (timing-only, wrong results): they exist
// in -DFASTMAX_ABLATIONS builds only
static bool wrong_result_variant(int v) { return v == 119 || v == 129 || v == 219 || v == 201 || (v >= 204 && v <= 209); }
namespace {
struct TuneEntry { const char* name; const char* env; int value; };
TuneEntry g_tune[TUNE_COUNT] = {
    {"mfma_variant", "FASTMAX_MFMA_VARIANT", 200},    // headline forward kernel: 200 = second generation (fastmax_mfma_v2.hip)
    {"bf16_kernel", "FASTMAX_BF16_KERNEL", 1},
    {"gemm_sched", "FASTMAX_GEMM_SCHED", 0},          // QLoRA GEMM: vector instructions per matrix instruction in the decode steps
    {"gemm_group_m", "FASTMAX_GEMM_GROUP_M", 16},     // QLoRA / head GEMM: row blocks per group of the workgroup -> tile map (0: column blocks fastest over the whole matrix)
    {"gemm_xcd", "FASTMAX_GEMM_XCD", 1},              // QLoRA / head GEMM tile map: 1 = every XCD owns a contiguous run of tiles, in 8-row-block groups
};
bool g_tune_loaded = false;
void tune_load() {
    if (g_tune_loaded) return;
    for (int i = 0; i < TUNE_COUNT; ++i) {
        const char* e = getenv(g_tune[i].env);
        if (!e) continue;
        if (i == TUNE_BF16_KERNEL) g_tune[i].value = e[0] == 'g' ? 0 : 1;
        else g_tune[i].value = atoi(e);
#ifndef FASTMAX_ABLATIONS
        if (i == TUNE_MFMA_VARIANT && wrong_result_variant(g_tune[i].value)) g_tune[i].value = 200;   // not in this build
#endif
    }
    g_tune_loaded = true;
}
}  // namespace
int tune_get(int key) {
    tune_load();
    return g_tune[key].value;
}
int tune_set(const char* name, int value) {
    tune_load();
    for (int i = 0; i < TUNE_COUNT; ++i)
        if (!strcmp(name, g_tune[i].name)) {
#ifndef FASTMAX_ABLATIONS
            if (i == TUNE_MFMA_VARIANT && wrong_result_variant(value)) return FASTMAX_E_BAD_SHAPE;
#endif
            g_tune[i].value = value;
            return FASTMAX_OK;
        }
    return FASTMAX_E_BAD_SHAPE;
}
int tune_get_by_name(const char* name) {
    tune_load();
    for (int i = 0; i < TUNE_COUNT; ++i)
        if (!strcmp(name, g_tune[i].name)) return g_tune[i].value;
    return INT_MIN;
}
}  // namespace fastmax

extern "C" {

int fastmax_hip_abi_version(void) { return FASTMAX_ABI_VERSION; }

int fastmax_hip_tune(const char* name, int value) { return name ? tune_set(name, value) : FASTMAX_E_NULL; }
int fastmax_hip_tune_get(const char* name) { return name ? tune_get_by_name(name) : INT_MIN; }
int fastmax_hip_build_flags(void) {
#ifdef FASTMAX_ABLATIONS
    return 1;
#else
    return 0;
#endif
}

const char* fastmax_hip_error_string(int code) {
    switch (code) {
        case FASTMAX_OK: return "ok";
        case FASTMAX_E_BAD_P: return "p should be 1 or 2";
        case FASTMAX_E_BAD_SHAPE: return "bad shape (sizes must be positive, causal needs Nq == Nk, D <= 256) or path not applicable";
        case FASTMAX_E_BAD_DTYPE: return "bad dtype";
        case FASTMAX_E_WORKSPACE: return "workspace missing or too small";
        case FASTMAX_E_ALIGNMENT: return "pointer / stride alignment";
        case FASTMAX_E_NULL: return "null pointer";
    }
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "unknown error";
}

int fastmax_hip_select_path(const fastmax_problem* prob) {
    const int rc = validate(prob);
    if (rc) return rc;
    return select(*prob);
}

size_t fastmax_hip_forward_workspace(const fastmax_problem* prob) {
    if (validate(prob)) return 0;
    if (select(*prob) != FASTMAX_PATH_MFMA) return 0;
    if (!prob->causal) return unmasked_lin_workspace(*prob);
    return split_workspace_bytes(*prob, prob->D <= 64 ? 64 : 128);
}

int fastmax_hip_forward(const fastmax_problem* prob, const void* q, const int64_t* q_strides, const void* k,
                        const int64_t* k_strides, const void* v, const int64_t* v_strides, void* o, float* g,
                        void* workspace, size_t workspace_bytes, void* stream) {
    int rc = validate(prob);
    if (rc) return rc;
    if (!q || !k || !v || !o || !q_strides || !k_strides || !v_strides) return FASTMAX_E_NULL;
    int path = select(*prob);
    if (path < 0) return path;
    if (path == FASTMAX_PATH_MFMA || path == FASTMAX_PATH_QUADRATIC_MFMA) {
        const bool ok = aligned16(q, q_strides, prob->in_dtype) && aligned16(k, k_strides, prob->in_dtype) &&
                        aligned16(v, v_strides, prob->in_dtype) && !(reinterpret_cast<uintptr_t>(o) & 15);
        if (!ok) {
            if (prob->path == path) return FASTMAX_E_ALIGNMENT;          // the caller forced this family
            path = path == FASTMAX_PATH_MFMA ? (prob->causal ? FASTMAX_PATH_RECURRENT : FASTMAX_PATH_QUADRATIC) : FASTMAX_PATH_QUADRATIC;
        }
    }
    FwdArgs a{*prob, q, k, v, st(q_strides), st(k_strides), st(v_strides), o, g, workspace, workspace_bytes,
              reinterpret_cast<hipStream_t>(stream)};
    switch (path) {
        case FASTMAX_PATH_MFMA:
            if (!prob->causal) return launch_fwd_unmasked_p1(a);
            if (mfma_p1_supported(*prob)) return launch_fwd_mfma_p1(a);
            if (mfma_d128_2p_supported(*prob)) return launch_fwd_mfma_d128_2p(a, nullptr, nullptr);
            return use_bf16_kernel(*prob) ? launch_fwd_mfma_bf16(a, nullptr, nullptr) : launch_fwd_mfma_gen(a, nullptr, nullptr);
        case FASTMAX_PATH_RECURRENT: return launch_fwd_recurrent_p1(a);
        case FASTMAX_PATH_QUADRATIC_MFMA: return quad32_supported(a.prob) ? launch_fwd_quad32(a) : launch_fwd_quad_mfma(a);
        default: return launch_fwd_quadratic(a);
    }
}

size_t fastmax_hip_backward_workspace(const fastmax_problem* prob) {
    if (validate(prob)) return 0;
    const size_t a = bwd_quadratic_workspace(*prob), b = lin_bwd_workspace(*prob);
    size_t c = scan_bwd_supported(*prob) ? scan_bwd_workspace(*prob) : 0;
    const size_t d = unmasked_lin_bwd_workspace(*prob);
    if (d > c) c = d;
    return a > b ? (a > c ? a : c) : (b > c ? b : c);
}

size_t fastmax_hip_forward_state_bytes(const fastmax_problem* prob, const void* q, const int64_t* q_strides, const void* k,
                                       const int64_t* k_strides, const void* v, const int64_t* v_strides, const void* o) {
    if (validate(prob) || !q || !k || !v || !o || !q_strides || !k_strides || !v_strides) return 0;
    if (select(*prob) != FASTMAX_PATH_MFMA || !prob->causal) return 0;
    // the same layout rule as fastmax_hip_forward: anything else takes a kernel without a sequence split
    if (!(aligned16(q, q_strides, prob->in_dtype) && aligned16(k, k_strides, prob->in_dtype) && aligned16(v, v_strides, prob->in_dtype) &&
          !(reinterpret_cast<uintptr_t>(o) & 15)))
        return 0;
    if (split_plan(*prob).nseg <= 1) return 0;
    return split_workspace_bytes(*prob, prob->D <= 64 ? 64 : 128);
}

int fastmax_hip_backward_with_states(const fastmax_problem* prob, const void* q, const int64_t* q_strides, const void* k,
                         const int64_t* k_strides, const void* v, const int64_t* v_strides, const void* o,
                         const float* g, const void* grad_o, const int64_t* go_strides, void* dq, void* dk, void* dv,
                         void* workspace, size_t workspace_bytes, const void* fwd_states, size_t fwd_state_bytes, void* stream) {
    int rc = validate(prob);
    if (rc) return rc;
    if (!q || !k || !v || !o || !g || !grad_o || !dq || !dk || !dv || !q_strides || !k_strides || !v_strides ||
        !go_strides)
        return FASTMAX_E_NULL;
    BwdArgs a{*prob, q, k, v, o, grad_o, g, st(q_strides), st(k_strides), st(v_strides), st(go_strides), dq, dk, dv,
              workspace, workspace_bytes, reinterpret_cast<hipStream_t>(stream)};
    if (fwd_states && fwd_state_bytes >= fastmax_hip_forward_state_bytes(prob, q, q_strides, k, k_strides, v, v_strides, o) && fwd_state_bytes > 0 &&
        !(reinterpret_cast<uintptr_t>(fwd_states) & 15))
        a.fwd_states = reinterpret_cast<const float*>(fwd_states);
    // matrix-core tiles unless the caller forces the vector-ALU family or the layout rules it out
    const bool mfma_ok = prob->path != FASTMAX_PATH_QUADRATIC && quad_mfma_bwd_supported(*prob) &&
                         aligned16(q, q_strides, prob->in_dtype) && aligned16(k, k_strides, prob->in_dtype) &&
                         aligned16(v, v_strides, prob->in_dtype) && aligned16(grad_o, go_strides, prob->in_dtype) &&
                         !((reinterpret_cast<uintptr_t>(dq) | reinterpret_cast<uintptr_t>(dk) |
                            reinterpret_cast<uintptr_t>(dv)) & 15);
    if (!mfma_ok) return launch_bwd_quadratic(a);
    // p=1 unmasked at sizes where totals + row-wise D x D products beat the O(N_q N_k) tiles
    if (prob->path != FASTMAX_PATH_QUADRATIC_MFMA && unmasked_lin_bwd_supported(*prob) && !(reinterpret_cast<uintptr_t>(o) & 15))
        return launch_bwd_unmasked_p1(a);
    // p=1 masked: linear-time scans (carried D x D state) unless the caller asks for the tile kernels
    const bool lin = prob->path != FASTMAX_PATH_QUADRATIC_MFMA && lin_bwd_supported(*prob) && prob->in_dtype == prob->out_dtype &&
                     !(reinterpret_cast<uintptr_t>(o) & 15);
    if (lin) return launch_bwd_lin(a);
    // fp32 / fp16 at 64 < D <= 128: the same scans with two-part operands, one per gradient (fastmax_scan_d128_2p.hip)
    if (prob->path != FASTMAX_PATH_QUADRATIC_MFMA && scan_bwd_supported(*prob) && !(reinterpret_cast<uintptr_t>(o) & 15))
        return launch_bwd_scan(a);
    return quad32_bwd_supported(*prob) ? launch_bwd_quad32(a) : launch_bwd_quad_mfma(a);
}

int fastmax_hip_backward(const fastmax_problem* prob, const void* q, const int64_t* q_strides, const void* k,
                         const int64_t* k_strides, const void* v, const int64_t* v_strides, const void* o,
                         const float* g, const void* grad_o, const int64_t* go_strides, void* dq, void* dk, void* dv,
                         void* workspace, size_t workspace_bytes, void* stream) {
    return fastmax_hip_backward_with_states(prob, q, q_strides, k, k_strides, v, v_strides, o, g, grad_o, go_strides, dq, dk, dv, workspace,
                                            workspace_bytes, nullptr, 0, stream);
}

size_t fastmax_hip_normalize_workspace(int B, int H) { return sizeof(unsigned int) * (size_t)B * H; }

int fastmax_hip_normalize(const void* x, const int64_t* x_strides, int dtype, float* y, float* inv_norm, int B, int H,
                          int N, int D, void* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !x_strides || !y) return FASTMAX_E_NULL;
    if (B <= 0 || H <= 0 || N <= 0 || D <= 0 || D > FASTMAX_MAX_D) return FASTMAX_E_BAD_SHAPE;
    if (!workspace || workspace_bytes < fastmax_hip_normalize_workspace(B, H)) return FASTMAX_E_WORKSPACE;
    return launch_normalize(x, st(x_strides), dtype, y, inv_norm, B, H, N, D, workspace,
                            reinterpret_cast<hipStream_t>(stream));
}

int fastmax_hip_normalize_stats(const void* x, const int64_t* x_strides, int dtype, float* inv_norm, int B, int H, int N,
                                int D, void* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !x_strides || !inv_norm) return FASTMAX_E_NULL;
    if (B <= 0 || H <= 0 || N <= 0 || D <= 0 || D > FASTMAX_MAX_D) return FASTMAX_E_BAD_SHAPE;
    if (!workspace || workspace_bytes < fastmax_hip_normalize_workspace(B, H)) return FASTMAX_E_WORKSPACE;
    return launch_normalize_stats(x, st(x_strides), dtype, inv_norm, B, H, N, D, workspace,
                                  reinterpret_cast<hipStream_t>(stream));
}

size_t fastmax_hip_normalize_stats2_workspace(int B, int H, int N) {
    return sizeof(unsigned long long) * 2 * (size_t)B * H * (size_t)((N + 255) / 256);          // (value, row) keys
}

int fastmax_hip_normalize_stats2(const void* x0, const int64_t* x0_strides, const void* x1, const int64_t* x1_strides, int dtype,
                                 float* inv_norm0, float* inv_norm1, int B, int H, int N, int D, void* workspace,
                                 size_t workspace_bytes, void* stream) {
    if (!x0 || !x1 || !x0_strides || !x1_strides || !inv_norm0 || !inv_norm1) return FASTMAX_E_NULL;
    if (B <= 0 || H <= 0 || N <= 0 || D <= 0 || D > FASTMAX_MAX_D || (int64_t)B * H > 65535) return FASTMAX_E_BAD_SHAPE;
    if (!workspace || workspace_bytes < fastmax_hip_normalize_stats2_workspace(B, H, N)) return FASTMAX_E_WORKSPACE;
    return launch_normalize_stats2(x0, st(x0_strides), x1, st(x1_strides), dtype, inv_norm0, inv_norm1, B, H, N, D, workspace,
                                   reinterpret_cast<hipStream_t>(stream));
}

int fastmax_hip_normalize_cast(const void* x, const int64_t* x_strides, int dtype, void* y, float* inv_norm, int B, int H,
                               int N, int D, void* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !x_strides || !y || !inv_norm) return FASTMAX_E_NULL;
    if (B <= 0 || H <= 0 || N <= 0 || D <= 0 || D > FASTMAX_MAX_D) return FASTMAX_E_BAD_SHAPE;
    if (!workspace || workspace_bytes < fastmax_hip_normalize_workspace(B, H)) return FASTMAX_E_WORKSPACE;
    const int es = dtype == FASTMAX_F32 ? 4 : 2;
    if (dtype < 0 || dtype > FASTMAX_F16) return FASTMAX_E_BAD_DTYPE;
    if (D * es > 1024) return FASTMAX_E_BAD_SHAPE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int npart = (N + 255) / 256;
    if (workspace_bytes >= sizeof(unsigned int) * (size_t)B * H * npart) {
        // two launches: per-block maxima, then the row pass combines them (no zeroing pass, no atomics, no finish pass)
        unsigned int* partials = reinterpret_cast<unsigned int*>(workspace);
        int rc = launch_normalize_partial_max(x, st(x_strides), dtype, partials, B, H, N, D, s);
        if (rc) return rc;
        return launch_normalize_cast(x, st(x_strides), dtype, y, nullptr, B, H, N, D, s, partials, npart, inv_norm);
    }
    int rc = launch_normalize_stats(x, st(x_strides), dtype, inv_norm, B, H, N, D, workspace, s);
    if (rc) return rc;
    return launch_normalize_cast(x, st(x_strides), dtype, y, inv_norm, B, H, N, D, s);
}

size_t fastmax_hip_normalize_backward_workspace(int B, int H, int N) { return normalize_backward_workspace(B, H, N); }

int fastmax_hip_normalize_backward(const void* x, const int64_t* x_strides, int dtype, const void* grad_y, const float* inv_norm,
                                   void* grad_x, int B, int H, int N, int D, void* workspace, size_t workspace_bytes,
                                   void* stream) {
    if (!x || !x_strides || !grad_y || !inv_norm || !grad_x) return FASTMAX_E_NULL;
    if (B <= 0 || H <= 0 || N <= 0 || D <= 0 || D > FASTMAX_MAX_D) return FASTMAX_E_BAD_SHAPE;
    if (!workspace || workspace_bytes < normalize_backward_workspace(B, H, N)) return FASTMAX_E_WORKSPACE;
    return launch_normalize_backward(x, st(x_strides), dtype, grad_y, inv_norm, grad_x, B, H, N, D, workspace,
                                     reinterpret_cast<hipStream_t>(stream));
}

// grouped-query form of the two entry points above: x holds the G key heads, y / grad_y the G * rep query-head copies
// (head g * rep + j) that the attention reads -- the GQA expand of lit_gpt/model.py:404-411 fused into the prologue's store,
// and the sum over a group's heads fused into its backward
int fastmax_hip_normalize_cast_expand(const void* x, const int64_t* x_strides, int dtype, void* y, float* inv_norm, int B, int G,
                                      int rep, int N, int D, void* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !x_strides || !y || !inv_norm) return FASTMAX_E_NULL;
    if (B <= 0 || G <= 0 || rep <= 0 || N <= 0 || D <= 0 || D > FASTMAX_MAX_D) return FASTMAX_E_BAD_SHAPE;
    if (dtype < 0 || dtype > FASTMAX_F16) return FASTMAX_E_BAD_DTYPE;
    const int es = dtype == FASTMAX_F32 ? 4 : 2;
    if (D * es > 1024) return FASTMAX_E_BAD_SHAPE;
    const int npart = (N + 255) / 256;
    if (!workspace || workspace_bytes < sizeof(unsigned int) * (size_t)B * G * npart) return FASTMAX_E_WORKSPACE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    unsigned int* partials = reinterpret_cast<unsigned int*>(workspace);
    const int rc = launch_normalize_partial_max(x, st(x_strides), dtype, partials, B, G, N, D, s);
    if (rc) return rc;
    return launch_normalize_cast(x, st(x_strides), dtype, y, nullptr, B, G, N, D, s, partials, npart, inv_norm, rep);
}

int fastmax_hip_normalize_backward_expand(const void* x, const int64_t* x_strides, int dtype, const void* grad_y,
                                          const float* inv_norm, void* grad_x, int B, int G, int rep, int N, int D, void* workspace,
                                          size_t workspace_bytes, void* stream) {
    if (!x || !x_strides || !grad_y || !inv_norm || !grad_x) return FASTMAX_E_NULL;
    if (B <= 0 || G <= 0 || rep <= 0 || N <= 0 || D <= 0 || D > FASTMAX_MAX_D) return FASTMAX_E_BAD_SHAPE;
    if (!workspace || workspace_bytes < normalize_backward_workspace_grouped(B, G, rep, N)) return FASTMAX_E_WORKSPACE;
    return launch_normalize_backward(x, st(x_strides), dtype, grad_y, inv_norm, grad_x, B, G, N, D, workspace,
                                     reinterpret_cast<hipStream_t>(stream), rep);
}

int fastmax_hip_linearmax_forward(const fastmax_problem* prob, const void* q, const int64_t* q_strides, const void* k,
                                  const int64_t* k_strides, const void* v, const int64_t* v_strides,
                                  const float* q_inv_norm, const float* k_inv_norm, void* o, float* g, void* workspace,
                                  size_t workspace_bytes, void* stream) {
    int rc = validate(prob);
    if (rc) return rc;
    if (!q || !k || !v || !o || !q_strides || !k_strides || !v_strides || !q_inv_norm || !k_inv_norm) return FASTMAX_E_NULL;
    if (!mfma_gen_supported(*prob, true) && !mfma_d128_2p_supported(*prob)) return FASTMAX_E_BAD_SHAPE;
    if (!(aligned16(q, q_strides, prob->in_dtype) && aligned16(k, k_strides, prob->in_dtype) &&
          aligned16(v, v_strides, prob->in_dtype)) || (reinterpret_cast<uintptr_t>(o) & 15))
        return FASTMAX_E_ALIGNMENT;
    FwdArgs a{*prob, q, k, v, st(q_strides), st(k_strides), st(v_strides), o, g, workspace, workspace_bytes,
              reinterpret_cast<hipStream_t>(stream)};
    if (mfma_d128_2p_supported(*prob)) return launch_fwd_mfma_d128_2p(a, q_inv_norm, k_inv_norm);
    return use_bf16_kernel(*prob) ? launch_fwd_mfma_bf16(a, q_inv_norm, k_inv_norm) : launch_fwd_mfma_gen(a, q_inv_norm, k_inv_norm);
}

// fastmax_hack.py:36-60 (masked branch) in ONE call: statistics + scan.  With the sequence split the statistics ride on the
// split's state pass (K is read there anyway, the state is linear in K's scale; Q's words come from extra blocks of the same
// launch); otherwise they are the paired statistics pass.  q_inv_norm / k_inv_norm (B*H floats each) are OUTPUTS here.
// workspace = [forward workspace | statistic words].
static size_t linearmax_stats_bytes(int B, int H, int N) {
    const size_t per_head = (size_t)((N + 255) / 256) + 32;          // statistics-only blocks of 256 rows + one key per segment
    return sizeof(unsigned long long) * 2 * (size_t)B * H * per_head;
}
static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

size_t fastmax_hip_linearmax_forward_auto_workspace(const fastmax_problem* prob) {
    if (validate(prob)) return 0;
    return align256(fastmax_hip_forward_workspace(prob)) + linearmax_stats_bytes(prob->B, prob->H, prob->Nq);
}

int fastmax_hip_linearmax_forward_auto(const fastmax_problem* prob, const void* q, const int64_t* q_strides, const void* k,
                                       const int64_t* k_strides, const void* v, const int64_t* v_strides, float* q_inv_norm,
                                       float* k_inv_norm, int* q_nstar, int* k_nstar, void* o, float* g, void* workspace,
                                       size_t workspace_bytes, void* stream) {
    int rc = validate(prob);
    if (rc) return rc;
    if (!q || !k || !v || !o || !q_strides || !k_strides || !v_strides || !q_inv_norm || !k_inv_norm) return FASTMAX_E_NULL;
    if (!mfma_gen_supported(*prob, true) && !mfma_d128_2p_supported(*prob)) return FASTMAX_E_BAD_SHAPE;
    if ((int64_t)prob->B * prob->H > 65535) return FASTMAX_E_BAD_SHAPE;
    if (!(aligned16(q, q_strides, prob->in_dtype) && aligned16(k, k_strides, prob->in_dtype) &&
          aligned16(v, v_strides, prob->in_dtype)) || (reinterpret_cast<uintptr_t>(o) & 15))
        return FASTMAX_E_ALIGNMENT;
    const size_t fwd_bytes = align256(fastmax_hip_forward_workspace(prob));
    if (!workspace || workspace_bytes < fwd_bytes + linearmax_stats_bytes(prob->B, prob->H, prob->Nq)) return FASTMAX_E_WORKSPACE;
    const LinearmaxStats stats{q_inv_norm, k_inv_norm, reinterpret_cast<unsigned int*>(static_cast<char*>(workspace) + fwd_bytes),
                               q_nstar, k_nstar};
    FwdArgs a{*prob, q, k, v, st(q_strides), st(k_strides), st(v_strides), o, g, workspace, fwd_bytes,
              reinterpret_cast<hipStream_t>(stream), &stats};
    if (mfma_d128_2p_supported(*prob)) return launch_fwd_mfma_d128_2p(a, q_inv_norm, k_inv_norm);
    return use_bf16_kernel(*prob) ? launch_fwd_mfma_bf16(a, q_inv_norm, k_inv_norm) : launch_fwd_mfma_gen(a, q_inv_norm, k_inv_norm);
}

// Training route of the same branch: the backward of fastmax_hip_linearmax_forward_auto.  q, k are the RAW tensors and
// q_inv_norm / k_inv_norm what the forward left; the linear-time scans apply the prologue while staging (as the forward does), so
// no normalised copy of q or k is ever stored.  dq, dk are the gradients wrt the NORMALISED q, k: the caller finishes with
// fastmax_hip_normalize_backward(q, dq, q_inv_norm) / (k, dk, k_inv_norm) -- unless flags bit 0 is set: then the dK/dV kernel
// applies the prologue's backward to its dK tile itself and dk is the gradient wrt the raw k (one k head per query head only).  fwd_states = the forward's workspace (its prefix
// states), or null.  FASTMAX_E_BAD_SHAPE where the linear-time backward does not cover the problem (fastmax_hip_linearmax_train_supported).
int fastmax_hip_linearmax_train_supported(const fastmax_problem* prob) {
    if (validate(prob)) return 0;
    if (!(prob->p == 1 && prob->causal) || prob->in_dtype != prob->out_dtype) return 0;
    return (mfma_gen_supported(*prob, true) && lin_bwd_supported(*prob)) ? 1 : 0;
}

int fastmax_hip_linearmax_backward(const fastmax_problem* prob, const void* q, const int64_t* q_strides, const void* k,
                                   const int64_t* k_strides, const void* v, const int64_t* v_strides, const void* o, const float* g,
                                   const void* grad_o, const int64_t* go_strides, const float* q_inv_norm, const float* k_inv_norm,
                                   const int* q_nstar, const int* k_nstar, void* dq, void* dk, void* dv, void* workspace, size_t workspace_bytes,
                                   const void* fwd_states, size_t fwd_state_bytes, int flags, void* stream) {
    int rc = validate(prob);
    if (rc) return rc;
    if (!q || !k || !v || !o || !g || !grad_o || !dq || !dk || !dv || !q_strides || !k_strides || !v_strides || !go_strides ||
        !q_inv_norm || !k_inv_norm)
        return FASTMAX_E_NULL;
    if (!fastmax_hip_linearmax_train_supported(prob)) return FASTMAX_E_BAD_SHAPE;
    if (!(aligned16(q, q_strides, prob->in_dtype) && aligned16(k, k_strides, prob->in_dtype) && aligned16(v, v_strides, prob->in_dtype) &&
          aligned16(grad_o, go_strides, prob->in_dtype)) ||
        ((reinterpret_cast<uintptr_t>(dq) | reinterpret_cast<uintptr_t>(dk) | reinterpret_cast<uintptr_t>(dv) | reinterpret_cast<uintptr_t>(o)) & 15))
        return FASTMAX_E_ALIGNMENT;
    BwdArgs a{*prob, q, k, v, o, grad_o, g, st(q_strides), st(k_strides), st(v_strides), st(go_strides), dq, dk, dv,
              workspace, workspace_bytes, reinterpret_cast<hipStream_t>(stream)};
    a.qscale = q_inv_norm;
    a.kscale = k_inv_norm;
    a.fuse_prologue = ((flags & 1) && k_nstar ? 1 : 0) | ((flags & 2) && q_nstar ? 2 : 0);
    a.k_nstar = k_nstar;
    a.q_nstar = q_nstar;
    const SplitPlan plan = split_plan(*prob);
    if (fwd_states && plan.nseg > 1 && fwd_state_bytes >= split_workspace_bytes(*prob, prob->D <= 64 ? 64 : 128) &&
        !(reinterpret_cast<uintptr_t>(fwd_states) & 15))
        a.fwd_states = reinterpret_cast<const float*>(fwd_states);
    return launch_bwd_lin(a);
}

}  // extern "C"
