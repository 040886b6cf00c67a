/* CPU oracle (plain C) for the fastmax / linearmax path -- TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load the
 * library built from this file (oracle/Makefile -> oracle/_build/libfastmax_oracle.so).
 * The product (fastmax_experiments_amd/) never links or calls it.
 *
 * It restates, token by token with a carried state, what the reference computes with
 * einsum + cumsum over materialised outer products:
 *   forward   attention_mechanisms/fastmax.py:218-250 (F masked) 287-322 (g masked)
 *             184-216 / 252-285 (unmasked), 97 (o = F/g), 78-82 (normalize_term rule: the
 *             caller passes a = 1/nt, b = 1/(2 nt^2))
 *   backward  fastmax.py:432-485 (dQ), 541-604 (dK), 647-691 (dV) masked;
 *             383-430, 487-539, 606-645 unmasked
 * in the unified form, with v' = [v | 1]:
 *   [F|g]_i = S1 + a q_i^T S2 + b (q_i x q_i):S3,   S* = sums over j<=i (masked) / all j
 * Inputs are float32 (what the kernels see); all arithmetic and outputs are float64.
 * Parity of this file is pinned by tests/test_oracle_c.py against the golden vectors
 * generated from the reference (tests/golden/).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
    int D, p;
    double a, b;
    double *S1; /* [D+1]          */
    double *S2; /* [D][D+1]       */
    double *S3; /* [D][D][D+1]    */
} state_t;

static int state_init(state_t *s, int D, int p, double a, double b) {
    s->D = D; s->p = p; s->a = a; s->b = b;
    size_t R = (size_t)D + 1;
    s->S1 = (double *)calloc(R, sizeof(double));
    s->S2 = (double *)calloc((size_t)D * R, sizeof(double));
    s->S3 = p == 2 ? (double *)calloc((size_t)D * D * R, sizeof(double)) : NULL;
    return s->S1 && s->S2 && (p != 2 || s->S3);
}
static void state_zero(state_t *s) {
    size_t R = (size_t)s->D + 1;
    memset(s->S1, 0, R * sizeof(double));
    memset(s->S2, 0, (size_t)s->D * R * sizeof(double));
    if (s->S3) memset(s->S3, 0, (size_t)s->D * s->D * R * sizeof(double));
}
static void state_free(state_t *s) { free(s->S1); free(s->S2); free(s->S3); }

/* S1 += y ; S2 += x (x) y ; S3 += x (x) x (x) y      (x: D, y: D+1) */
static void state_update(state_t *s, const double *x, const double *y) {
    const int D = s->D, R = D + 1;
    for (int r = 0; r < R; ++r) s->S1[r] += y[r];
    for (int m = 0; m < D; ++m) {
        double *row = s->S2 + (size_t)m * R;
        const double xm = x[m];
        for (int r = 0; r < R; ++r) row[r] += xm * y[r];
    }
    if (s->p == 2)
        for (int m = 0; m < D; ++m)
            for (int l = 0; l < D; ++l) {
                double *row = s->S3 + ((size_t)m * D + l) * R;
                const double xx = x[m] * x[l];
                for (int r = 0; r < R; ++r) row[r] += xx * y[r];
            }
}

/* out[r] = S1[r] + a sum_m x_m S2[m][r] + b sum_ml x_m x_l S3[m][l][r]    (r < D+1) */
static void readout_r(const state_t *s, const double *x, double *out) {
    const int D = s->D, R = D + 1;
    for (int r = 0; r < R; ++r) out[r] = s->S1[r];
    for (int m = 0; m < D; ++m) {
        const double *row = s->S2 + (size_t)m * R;
        const double c = s->a * x[m];
        for (int r = 0; r < R; ++r) out[r] += c * row[r];
    }
    if (s->p == 2)
        for (int m = 0; m < D; ++m)
            for (int l = 0; l < D; ++l) {
                const double *row = s->S3 + ((size_t)m * D + l) * R;
                const double c = s->b * x[m] * x[l];
                for (int r = 0; r < R; ++r) out[r] += c * row[r];
            }
}

/* out[m] = a sum_r S2[m][r] y_r + 2b sum_l x_l sum_r S3[m][l][r] y_r      (m < D) */
static void readout_m(const state_t *s, const double *x, const double *y, double *out) {
    const int D = s->D, R = D + 1;
    for (int m = 0; m < D; ++m) {
        const double *row = s->S2 + (size_t)m * R;
        double acc = 0.0;
        for (int r = 0; r < R; ++r) acc += row[r] * y[r];
        out[m] = s->a * acc;
    }
    if (s->p == 2)
        for (int m = 0; m < D; ++m) {
            double acc = 0.0;
            for (int l = 0; l < D; ++l) {
                const double *row = s->S3 + ((size_t)m * D + l) * R;
                double t = 0.0;
                for (int r = 0; r < R; ++r) t += row[r] * y[r];
                acc += x[l] * t;
            }
            out[m] += 2.0 * s->b * acc;
        }
}

static void load_row(const float *src, int D, double *dst) {
    for (int d = 0; d < D; ++d) dst[d] = (double)src[d];
}

/* Forward.  q:(B,H,Nq,D) k,v:(B,H,Nk,D) contiguous float32; o:(B,H,Nq,D), g:(B,H,Nq)
 * float64.  causal: 1 = masked (needs Nq==Nk).  g0: constant term of the denominator in
 * the unmasked case (fastmax.py:271 uses Nq, fastmax_hack.py:21 uses Nk).
 * Returns 0, or -1 on bad arguments / allocation failure. */
int fastmax_oracle_fwd(const float *q, const float *k, const float *v, double *o, double *g,
                       int B, int H, int Nq, int Nk, int D, int p, int causal,
                       double a, double b, double g0, int nthreads) {
    if ((p != 1 && p != 2) || D <= 0 || (causal && Nq != Nk)) return -1;
    const int R = D + 1;
    int fail = 0;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(dynamic, 1)
    for (int bh = 0; bh < B * H; ++bh) {
        state_t st;
        double *x = (double *)malloc(sizeof(double) * (size_t)(3 * R));
        if (!x || !state_init(&st, D, p, a, b)) { fail = 1; free(x); continue; }
        double *y = x + R, *out = x + 2 * R;
        const float *qh = q + (size_t)bh * Nq * D, *kh = k + (size_t)bh * Nk * D,
                    *vh = v + (size_t)bh * Nk * D;
        double *oh = o + (size_t)bh * Nq * D, *gh = g + (size_t)bh * Nq;
        if (!causal)
            for (int j = 0; j < Nk; ++j) {
                load_row(kh + (size_t)j * D, D, x);
                load_row(vh + (size_t)j * D, D, y); y[D] = 1.0;
                state_update(&st, x, y);
            }
        for (int i = 0; i < Nq; ++i) {
            if (causal) {
                load_row(kh + (size_t)i * D, D, x);
                load_row(vh + (size_t)i * D, D, y); y[D] = 1.0;
                state_update(&st, x, y);
            }
            load_row(qh + (size_t)i * D, D, x);
            readout_r(&st, x, out);
            double gi = out[D];
            if (!causal) gi += g0 - (double)Nk;
            gh[i] = gi;
            for (int d = 0; d < D; ++d) oh[(size_t)i * D + d] = out[d] / gi;
        }
        state_free(&st); free(x);
    }
    return fail ? -1 : 0;
}

/* Backward.  grad:(B,H,Nq,D) float32 upstream dL/do; o,g: the forward's float64 outputs.
 * dq:(B,H,Nq,D) dk,dv:(B,H,Nk,D) float64. */
int fastmax_oracle_bwd(const float *q, const float *k, const float *v, const double *o,
                       const double *g, const float *grad, double *dq, double *dk, double *dv,
                       int B, int H, int Nq, int Nk, int D, int p, int causal,
                       double a, double b, int nthreads) {
    if ((p != 1 && p != 2) || D <= 0 || (causal && Nq != Nk)) return -1;
    const int R = D + 1;
    int fail = 0;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(dynamic, 1)
    for (int bh = 0; bh < B * H; ++bh) {
        state_t st;
        double *x = (double *)malloc(sizeof(double) * (size_t)(3 * R));
        double *gh_all = (double *)malloc(sizeof(double) * (size_t)Nq * R);  /* Ghat rows */
        if (!x || !gh_all || !state_init(&st, D, p, a, b)) { fail = 1; free(x); free(gh_all); continue; }
        double *y = x + R, *out = x + 2 * R;
        const float *qh = q + (size_t)bh * Nq * D, *kh = k + (size_t)bh * Nk * D,
                    *vh = v + (size_t)bh * Nk * D, *Gh = grad + (size_t)bh * Nq * D;
        const double *oh = o + (size_t)bh * Nq * D, *gg = g + (size_t)bh * Nq;
        double *dqh = dq + (size_t)bh * Nq * D, *dkh = dk + (size_t)bh * Nk * D,
               *dvh = dv + (size_t)bh * Nk * D;
        /* Ghat_i = [G_i | -G_i.o_i] / g_i */
        for (int i = 0; i < Nq; ++i) {
            double c = 0.0, w = 1.0 / gg[i];
            for (int d = 0; d < D; ++d) c += (double)Gh[(size_t)i * D + d] * oh[(size_t)i * D + d];
            for (int d = 0; d < D; ++d) gh_all[(size_t)i * R + d] = w * (double)Gh[(size_t)i * D + d];
            gh_all[(size_t)i * R + D] = -w * c;
        }
        /* dQ: prefix state over keys */
        if (!causal)
            for (int j = 0; j < Nk; ++j) {
                load_row(kh + (size_t)j * D, D, x);
                load_row(vh + (size_t)j * D, D, y); y[D] = 1.0;
                state_update(&st, x, y);
            }
        for (int i = 0; i < Nq; ++i) {
            if (causal) {
                load_row(kh + (size_t)i * D, D, x);
                load_row(vh + (size_t)i * D, D, y); y[D] = 1.0;
                state_update(&st, x, y);
            }
            load_row(qh + (size_t)i * D, D, x);
            readout_m(&st, x, gh_all + (size_t)i * R, dqh + (size_t)i * D);
        }
        /* dK, dV: suffix state over queries (reverse scan) */
        state_zero(&st);
        if (!causal)
            for (int i = 0; i < Nq; ++i) {
                load_row(qh + (size_t)i * D, D, x);
                state_update(&st, x, gh_all + (size_t)i * R);
            }
        for (int j = Nk - 1; j >= 0; --j) {
            if (causal) {
                load_row(qh + (size_t)j * D, D, x);
                state_update(&st, x, gh_all + (size_t)j * R);
            }
            load_row(kh + (size_t)j * D, D, x);
            load_row(vh + (size_t)j * D, D, y); y[D] = 1.0;
            readout_m(&st, x, y, dkh + (size_t)j * D);
            readout_r(&st, x, out);
            for (int d = 0; d < D; ++d) dvh[(size_t)j * D + d] = out[d];
        }
        state_free(&st); free(x); free(gh_all);
    }
    return fail ? -1 : 0;
}

/* linearmax prologue, fastmax_hack.py:38-43 (== fastmax.py:326-334): per token subtract
 * the mean over D, then divide the whole (b,h) slab by the max over tokens of the
 * per-token L2 norm.  x:(BH,N,D) float32 in, float32 out (rounded once at the end). */
int fastmax_oracle_normalize(const float *x, float *y, int BH, int N, int D) {
    for (int bh = 0; bh < BH; ++bh) {
        const float *xs = x + (size_t)bh * N * D;
        float *ys = y + (size_t)bh * N * D;
        double mx = 0.0;
        for (int n = 0; n < N; ++n) {
            double mean = 0.0, nn = 0.0;
            for (int d = 0; d < D; ++d) mean += xs[(size_t)n * D + d];
            mean /= D;
            for (int d = 0; d < D; ++d) { double c = xs[(size_t)n * D + d] - mean; nn += c * c; }
            if (nn > mx) mx = nn;
        }
        const double inv = 1.0 / __builtin_sqrt(mx);
        for (int n = 0; n < N; ++n) {
            double mean = 0.0;
            for (int d = 0; d < D; ++d) mean += xs[(size_t)n * D + d];
            mean /= D;
            for (int d = 0; d < D; ++d) ys[(size_t)n * D + d] = (float)((xs[(size_t)n * D + d] - mean) * inv);
        }
    }
    return 0;
}

int fastmax_oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
