// fastmax p=1 masked forward on the CDNA4 matrix cores (gfx950), linear in N.
//
// One workgroup (4 waves) walks one (b,h) head in chunks of C = 64 tokens, carrying
//     S2 = sum_{j<chunk} k_j v_j^T   (D x D, fp32, in MFMA accumulators; a bf16 hi/lo image of it in LDS)
//     S1 = sum v_j,  ksum = sum k_j   (fp32, exact vector-ALU sums, in LDS)
// and computes per chunk, with q' = a*q (a = 1/nt):
//     O^T[d][i]  = S1[d] + (S2^T q'_i)[d]                  inter-chunk   (3)
//                + sum_{j<=i} (1 + q'_i.k_j) v_j[d]        intra-chunk   (1) S^T = K Q'^T, (2) O^T += V^T P^T
//     g_i        = (i+1) + q'_i.ksum_prev + sum_{j<=i in chunk} q'_i.k_j
//     S2        += K^T V                                    state update  (4)
// which is attention_mechanisms/fastmax.py:236-241 (F) and 306-312 (g) without the (N,D,D) temporaries.
//
// Numerics: fp32 inputs are split into bf16 hi + bf16 lo and every product uses 3 MFMAs
// (hi*hi + lo*hi + hi*lo, fp32 accumulate): ~2^-16 relative per product.  gfx950 has no xf32 and its
// fp32-input MFMA runs at 1/16 of the bf16 rate, which would leave this kernel matrix-bound.
//
// Work split: wave w owns queries 16w..16w+15 of the chunk (all D output columns) for (1)-(3) and the
// value-column slab 16w..16w+15 of S2 for (4).  All operands are MFMA 16x16x32 bf16 fragments:
//   * Q', K row reads (ds_read_b128) from row-major [token][m] images,
//   * V^T, K^T transposed reads (ds_read_b64_tr_b16) from the same row-major images,
//   * P^T and S2 come straight from accumulator registers (k order permuted to match, see kperm()).
// LDS images use 128-byte rows with the 16-byte chunk index XOR-ed with (row & 7): conflict-free for
// both kinds of read.
#include "fastmax_mfma_common.h"

#include <cstdlib>

namespace fastmax {

struct MfmaParams {
    const void *q, *k, *v;
    Strides3 qs, ks, vs;
    float* o;
    float* g;
    int H, N;
    float a;
    const float* state;   // sequence split: inclusive prefix states [(bh*(nseg-1) + seg-1)][64*64 + 64 + 64], or null
    int nseg, cps;        // segments per head, chunks per segment
};

namespace m64 {
constexpr int D = 64, C = 64, NT = 256;
constexpr int IMG = C * D * 2;               // one bf16 image: 8 KiB
constexpr int QH = 0, QL = IMG, KH = 2 * IMG, KL = 3 * IMG, VH = 4 * IMG, VL = 5 * IMG;
constexpr int S2H = 6 * IMG, S2L = 7 * IMG;  // [d][m] images of S2^T
constexpr int S1V = 8 * IMG;                 // 2 x 64 floats (double-buffered by chunk parity)
constexpr int KSUM = S1V + 512;              // 2 x 64 floats
constexpr int PARTV = KSUM + 512;            // 16 x 64 floats: per-row-group column sums of V
constexpr int PARTK = PARTV + 4096;
constexpr int QK = PARTK + 4096;             // 64 floats: q'_i . ksum_prev
constexpr int LDS_BYTES = QK + 256;          // 75008
}  // namespace m64

// ------------------------------------------------------------------------------------------------
// grid = B*H workgroups, block = 256 threads, dynamic LDS = m64::LDS_BYTES.  float32 I/O, D = 64.
// ------------------------------------------------------------------------------------------------
template <int PF, int ST, int SCHED>
__global__ __launch_bounds__(256, 2) void fwd_p1_mfma_d64_f32_kernel(MfmaParams prm) {
    using namespace m64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave id, provably uniform
    const int r = lane & 15, q4 = lane >> 4;
    const int bh = blockIdx.x / prm.nseg, seg = blockIdx.x - bh * prm.nseg;
    const int b = bh / prm.H, h = bh % prm.H;
    const int N = prm.N;
    const float a = prm.a;

    const float* qb = reinterpret_cast<const float*>(prm.q) + (int64_t)b * prm.qs.sb + (int64_t)h * prm.qs.sh;
    const float* kb = reinterpret_cast<const float*>(prm.k) + (int64_t)b * prm.ks.sb + (int64_t)h * prm.ks.sh;
    const float* vb = reinterpret_cast<const float*>(prm.v) + (int64_t)b * prm.vs.sb + (int64_t)h * prm.vs.sh;
    float* ob = prm.o + (int64_t)bh * N * D;
    float* gb = prm.g ? prm.g + (int64_t)bh * N : nullptr;

    // staging map: thread -> (row srow + 16u, float4 column scol)
    const int srow = tid >> 4, scol = tid & 15;
    auto issue_loads = [&](f32x4 (&rq)[4], f32x4 (&rk)[4], f32x4 (&rv)[4], int n0) {
        if (n0 + C <= N) {                                           // full chunk (block-uniform)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t gn = n0 + srow + 16 * u;
                if constexpr (ST == 2) {                             // streamed once: non-temporal policy
                    rq[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(qb + gn * prm.qs.sn + 4 * scol));
                    rk[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(kb + gn * prm.ks.sn + 4 * scol));
                    rv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(vb + gn * prm.vs.sn + 4 * scol));
                } else {
                    rq[u] = *reinterpret_cast<const f32x4*>(qb + gn * prm.qs.sn + 4 * scol);
                    rk[u] = *reinterpret_cast<const f32x4*>(kb + gn * prm.ks.sn + 4 * scol);
                    rv[u] = *reinterpret_cast<const f32x4*>(vb + gn * prm.vs.sn + 4 * scol);
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int gn = n0 + srow + 16 * u;
                const int gc = gn < N ? gn : N - 1;                  // clamp: always a legal address
                const float keep = gn < N ? 1.0f : 0.0f;
                rq[u] = *reinterpret_cast<const f32x4*>(qb + (int64_t)gc * prm.qs.sn + 4 * scol) * keep;
                rk[u] = *reinterpret_cast<const f32x4*>(kb + (int64_t)gc * prm.ks.sn + 4 * scol) * keep;
                rv[u] = *reinterpret_cast<const f32x4*>(vb + (int64_t)gc * prm.vs.sn + 4 * scol) * keep;
            }
        }
    };

    const int nchunks = (N + C - 1) / C;
    const int c_begin = seg * prm.cps, c_end = min(nchunks, c_begin + prm.cps);
    f32x4 s2acc[4];                                                  // S2[16mt + 4q4 + reg][16w + r]
    auto publish_s2 = [&]() {                                        // accumulators -> bf16 hi/lo image rows d = 16w + r
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            bf16x4 hi, lo;
            split4(s2acc[mt], hi, lo);
            const int off = img_off<64>(16 * w + r, 2 * mt + (q4 >> 1)) + ((q4 & 1) << 3);
            *reinterpret_cast<bf16x4*>(smem + S2H + off) = hi;
            *reinterpret_cast<bf16x4*>(smem + S2L + off) = lo;
        }
    };
    if (seg == 0) {
        // zero the carried state: S2 images, S1V[0], KSUM[0]
        for (int i = tid; i < (2 * IMG) / 16; i += NT) *reinterpret_cast<f32x4*>(smem + S2H + 16 * i) = f32x4{0, 0, 0, 0};
        if (tid < 64) {
            reinterpret_cast<float*>(smem + S1V)[tid] = 0.f;
            reinterpret_cast<float*>(smem + KSUM)[tid] = 0.f;
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) s2acc[mt] = f32x4{0, 0, 0, 0};
    } else {
        // sequence split: start from the prefix state of all earlier segments
        const float* rec = prm.state + ((int64_t)bh * (prm.nseg - 1) + (seg - 1)) * (64 * 64 + 128);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int i = 0; i < 4; ++i) s2acc[mt][i] = rec[(16 * mt + 4 * q4 + i) * 64 + 16 * w + r];
        publish_s2();
        if (tid < 64) {
            reinterpret_cast<float*>(smem + S1V)[64 * (c_begin & 1) + tid] = rec[64 * 64 + tid];
            reinterpret_cast<float*>(smem + KSUM)[64 * (c_begin & 1) + tid] = rec[64 * 64 + 64 + tid];
        }
    }

    auto chunk_body = [&](f32x4 (&rq)[4], f32x4 (&rk)[4], f32x4 (&rv)[4], int c) {
        const int n0 = c * C;
        if constexpr (SCHED == 9) {
            // ABLATION (timing only, wrong results): the kernel's HBM access pattern with no LDS / MFMA work
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int gn = n0 + srow + 16 * u;
                if (gn < N) {
                    if constexpr (ST == 2) __builtin_nontemporal_store(rq[u] + rk[u] + rv[u], reinterpret_cast<f32x4*>(ob + (int64_t)gn * D + 4 * scol));
                    else *reinterpret_cast<f32x4*>(ob + (int64_t)gn * D + 4 * scol) = rq[u] + rk[u] + rv[u];
                }
            }
            if (c + PF < c_end) issue_loads(rq, rk, rv, n0 + PF * C);
            return;
        }
        const int cur = c & 1, nxt = cur ^ 1;
        const float* ksum_cur = reinterpret_cast<const float*>(smem + KSUM) + 64 * cur;
        const float* s1v_cur = reinterpret_cast<const float*>(smem + S1V) + 64 * cur;

        // ---- (a) registers -> bf16 hi/lo images; exact fp32 side sums -----------------------------
        {
            const f32x4 ks4 = *reinterpret_cast<const f32x4*>(ksum_cur + 4 * scol);
            f32x4 ck = {0, 0, 0, 0}, cv = {0, 0, 0, 0};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int row = srow + 16 * u;
                const int off = img_off<64>(row, scol >> 1) + ((scol & 1) << 3);
                bf16x4 hi, lo;
                const f32x4 xq = rq[u] * a;
                split4(xq, hi, lo);
                *reinterpret_cast<bf16x4*>(smem + QH + off) = hi;
                *reinterpret_cast<bf16x4*>(smem + QL + off) = lo;
                float part = xq[0] * ks4[0] + xq[1] * ks4[1] + xq[2] * ks4[2] + xq[3] * ks4[3];
                part = row16_sum_to_lane15(part);
                if (scol == 15) reinterpret_cast<float*>(smem + QK)[row] = part;
                split4(rk[u], hi, lo);
                *reinterpret_cast<bf16x4*>(smem + KH + off) = hi;
                *reinterpret_cast<bf16x4*>(smem + KL + off) = lo;
                split4(rv[u], hi, lo);
                *reinterpret_cast<bf16x4*>(smem + VH + off) = hi;
                *reinterpret_cast<bf16x4*>(smem + VL + off) = lo;
                ck += rk[u];
                cv += rv[u];
            }
            *reinterpret_cast<f32x4*>(smem + PARTK + (srow * 64 + 4 * scol) * 4) = ck;
            *reinterpret_cast<f32x4*>(smem + PARTV + (srow * 64 + 4 * scol) * 4) = cv;
        }
        if (c + PF < c_end) issue_loads(rq, rk, rv, n0 + PF * C);    // refill this register set: PF chunks ahead
        __syncthreads();                                             // B1: images + partial sums visible

        // running sums for the NEXT chunk (double-buffered, so readers of `cur` are undisturbed)
        if (tid < 128) {
            const int col = tid & 63;
            const float* part = reinterpret_cast<const float*>(smem + (tid < 64 ? PARTV : PARTK));
            float* base = reinterpret_cast<float*>(smem + (tid < 64 ? S1V : KSUM));
            float s = base[64 * cur + col];
#pragma unroll
            for (int g16 = 0; g16 < 16; ++g16) s += part[g16 * 64 + col];
            base[64 * nxt + col] = s;
        }

        // ---- phase A: 16 queries per wave.  SCHED: the wave that takes the heavy (late) query tile rotates with
        // the chunk, and every MFMA batch has its fragment loads issued one batch ahead (pinned by sched_barrier)
        const int wq = SCHED ? ((w + c) & 3) : w;                    // query tile of this wave in this chunk
        const int qi = 16 * wq + r;                                  // query row inside the chunk
        f32x4 oacc[4];
        float gsum = 0.f;
        bf16x8 ph[2], pl[2];
        if constexpr (SCHED == 0) {
        bf16x8 qh[2], ql[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            qh[ks] = ld_row8<64>(smem, QH, qi, 4 * ks + q4);
            ql[ks] = ld_row8<64>(smem, QL, qi, 4 * ks + q4);
        }
        // (3) inter-chunk: O^T = S1 + S2^T Q'^T      (A = S2^T image rows d, B = Q'^T)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            oacc[dt] = *reinterpret_cast<const f32x4*>(s1v_cur + 16 * dt + 4 * q4);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16x8 sh = ld_row8<64>(smem, S2H, 16 * dt + r, 4 * ks + q4);
                const bf16x8 sl = ld_row8<64>(smem, S2L, 16 * dt + r, 4 * ks + q4);
                oacc[dt] = mfma3(sh, sl, qh[ks], ql[ks], oacc[dt]);
            }
        }
        // (1) scores S^T[j][i] = k_j . q'_i for key tiles jt <= w; mask the diagonal tile; P = 1 + s
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            f32x4 pt[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int jt = 2 * s + e;
                f32x4 sc = {0, 0, 0, 0};
                if (jt <= w) {                                       // wave-uniform
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        const bf16x8 kh = ld_row8<64>(smem, KH, 16 * jt + r, 4 * ks + q4);
                        const bf16x8 kl = ld_row8<64>(smem, KL, 16 * jt + r, 4 * ks + q4);
                        sc = mfma3(kh, kl, qh[ks], ql[ks], sc);
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bool keep = (jt < w) || (jt == w && (4 * q4 + i) <= r);
                    const float sv = keep ? sc[i] : 0.f;
                    gsum += sv;
                    pt[e][i] = keep ? 1.0f + sv : 0.f;
                }
            }
            bf16x4 h0, l0, h1, l1;
            split4(pt[0], h0, l0);
            split4(pt[1], h1, l1);
            ph[s] = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
            pl[s] = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
        }
        // (2) intra-chunk: O^T += V^T P^T   (A = V^T by transposed reads, B = P^T from registers)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (2 * s <= w) {                                        // wave-uniform
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const bf16x8 vh = ld_tr8<64>(smem, VH, 32 * s, 16 * dt, lane);
                    const bf16x8 vl = ld_tr8<64>(smem, VL, 32 * s, 16 * dt, lane);
                    oacc[dt] = mfma3(vh, vl, ph[s], pl[s], oacc[dt]);
                }
            }
        }
        } else {
#define FM_SB() __builtin_amdgcn_sched_barrier(0)
            bf16x8 qh[2], ql[2];
            bf16x8 fa[2][2][2], fb[2][2][2];                         // two fragment batches: [tile][ks][hi/lo]
            auto ld_rows = [&](bf16x8 (&f)[2][2][2], int baseh, int basel, int row0) {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        f[t][ks][0] = ld_row8<64>(smem, baseh, row0 + 16 * t + r, 4 * ks + q4);
                        f[t][ks][1] = ld_row8<64>(smem, basel, row0 + 16 * t + r, 4 * ks + q4);
                    }
            };
            auto mm_rows = [&](const bf16x8 (&f)[2][2][2], int t, f32x4 acc) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) acc = mfma3(f[t][ks][0], f[t][ks][1], qh[ks], ql[ks], acc);
                return acc;
            };
            // L0: Q fragments, S2^T rows of d-tiles 0,1, S1 (accumulator init)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                qh[ks] = ld_row8<64>(smem, QH, qi, 4 * ks + q4);
                ql[ks] = ld_row8<64>(smem, QL, qi, 4 * ks + q4);
            }
            ld_rows(fa, S2H, S2L, 0);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) oacc[dt] = *reinterpret_cast<const f32x4*>(s1v_cur + 16 * dt + 4 * q4);
            FM_SB();
            ld_rows(fb, S2H, S2L, 32);                               // L1: d-tiles 2,3
            FM_SB();
            oacc[0] = mm_rows(fa, 0, oacc[0]);                       // (3) inter-chunk
            oacc[1] = mm_rows(fa, 1, oacc[1]);
            FM_SB();
            ld_rows(fa, KH, KL, 0);                                  // L2: K rows of key tiles 0,1
            FM_SB();
            oacc[2] = mm_rows(fb, 0, oacc[2]);
            oacc[3] = mm_rows(fb, 1, oacc[3]);
            FM_SB();
            ld_rows(fb, KH, KL, 32);                                 // L3: key tiles 2,3 (used when wq >= 2)
            FM_SB();
            // (1) scores for key tiles jt <= wq, masked diagonal, P = 1 + s, split hi/lo
            f32x4 sc[4];
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) sc[jt] = f32x4{0, 0, 0, 0};
            sc[0] = mm_rows(fa, 0, sc[0]);
            if (wq >= 1) sc[1] = mm_rows(fa, 1, sc[1]);
            FM_SB();
            bf16x8 va[4][2], vb[4][2];                               // V^T fragments [dt][hi/lo] for k-steps 0 / 1
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {                         // L4: V^T, keys 0..31
                va[dt][0] = ld_tr8<64>(smem, VH, 0, 16 * dt, lane);
                va[dt][1] = ld_tr8<64>(smem, VL, 0, 16 * dt, lane);
            }
            FM_SB();
            if (wq >= 2) {
                sc[2] = mm_rows(fb, 0, sc[2]);
                if (wq >= 3) sc[3] = mm_rows(fb, 1, sc[3]);
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                f32x4 pt[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int jt = 2 * s + e;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const bool keep = (jt < wq) || (jt == wq && (4 * q4 + i) <= r);
                        const float sv = keep ? sc[jt][i] : 0.f;
                        gsum += sv;
                        pt[e][i] = keep ? 1.0f + sv : 0.f;
                    }
                }
                bf16x4 h0, l0, h1, l1;
                split4(pt[0], h0, l0);
                split4(pt[1], h1, l1);
                ph[s] = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
                pl[s] = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
            }
            FM_SB();
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {                         // L5: V^T, keys 32..63
                vb[dt][0] = ld_tr8<64>(smem, VH, 32, 16 * dt, lane);
                vb[dt][1] = ld_tr8<64>(smem, VL, 32, 16 * dt, lane);
            }
            FM_SB();
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) oacc[dt] = mfma3(va[dt][0], va[dt][1], ph[0], pl[0], oacc[dt]);   // (2) intra-chunk
            FM_SB();
            if (wq >= 2) {
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) oacc[dt] = mfma3(vb[dt][0], vb[dt][1], ph[1], pl[1], oacc[dt]);
            }
            FM_SB();
        }
        // denominator: count + q'.ksum_prev + intra-chunk score sum (over the 4 k-groups of the lane's query)
        gsum += __shfl_xor(gsum, 16, 64);
        gsum += __shfl_xor(gsum, 32, 64);
        const int gi = n0 + qi;
        const float gval = (float)(gi + 1) + reinterpret_cast<const float*>(smem + QK)[qi] + gsum;
        const float ginv = 1.0f / gval;
        if constexpr (ST == 0) {
            if (gi < N) {
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
                    *reinterpret_cast<f32x4*>(ob + (int64_t)gi * D + 16 * dt + 4 * q4) = oacc[dt] * ginv;
            }
        } else {
            // stage the wave's 16 x 64 fp32 tile through its own (already consumed) Q'-image rows so that
            // every store instruction writes 4 whole 256-byte rows: cols 0..31 -> QH rows, 32..63 -> QL rows
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
                *reinterpret_cast<f32x4*>(smem + ((dt >> 1) ? QL : QH) + img_off<64>(qi, (dt & 1) * 4 + q4)) = oacc[dt] * ginv;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int rl = 4 * u + q4, c16 = r;                 // row inside the wave tile, 16-byte column
                const f32x4 val = *reinterpret_cast<const f32x4*>(smem + ((c16 >> 3) ? QL : QH) + img_off<64>(16 * wq + rl, c16 & 7));
                const int go = n0 + 16 * wq + rl;
                if (go < N) {
                    if constexpr (ST == 2) __builtin_nontemporal_store(val, reinterpret_cast<f32x4*>(ob + (int64_t)go * D + 4 * c16));
                    else *reinterpret_cast<f32x4*>(ob + (int64_t)go * D + 4 * c16) = val;
                }
            }
        }
        if (gi < N && gb && q4 == 0) gb[gi] = gval;

        // ---- phase B: S2[:, 16w..16w+15] += K^T V  (A = K^T, B = V, both by transposed reads) --------
        if constexpr (SCHED == 0) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const bf16x8 vh = ld_tr8<64>(smem, VH, 32 * s, 16 * w, lane);
            const bf16x8 vl = ld_tr8<64>(smem, VL, 32 * s, 16 * w, lane);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const bf16x8 kh = ld_tr8<64>(smem, KH, 32 * s, 16 * mt, lane);
                const bf16x8 kl = ld_tr8<64>(smem, KL, 32 * s, 16 * mt, lane);
                s2acc[mt] = mfma3(kh, kl, vh, vl, s2acc[mt]);
            }
        }
        } else {
            bf16x8 kt0[4][2], kt1[4][2], vt0[2], vt1[2];
            vt0[0] = ld_tr8<64>(smem, VH, 0, 16 * w, lane);
            vt0[1] = ld_tr8<64>(smem, VL, 0, 16 * w, lane);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                kt0[mt][0] = ld_tr8<64>(smem, KH, 0, 16 * mt, lane);
                kt0[mt][1] = ld_tr8<64>(smem, KL, 0, 16 * mt, lane);
            }
            FM_SB();
            vt1[0] = ld_tr8<64>(smem, VH, 32, 16 * w, lane);
            vt1[1] = ld_tr8<64>(smem, VL, 32, 16 * w, lane);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                kt1[mt][0] = ld_tr8<64>(smem, KH, 32, 16 * mt, lane);
                kt1[mt][1] = ld_tr8<64>(smem, KL, 32, 16 * mt, lane);
            }
            FM_SB();
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) s2acc[mt] = mfma3(kt0[mt][0], kt0[mt][1], vt0[0], vt0[1], s2acc[mt]);
            FM_SB();
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) s2acc[mt] = mfma3(kt1[mt][0], kt1[mt][1], vt1[0], vt1[1], s2acc[mt]);
            FM_SB();
#undef FM_SB
        }
        __syncthreads();                                             // B2: every read of this chunk's images is done

        // ---- (e) publish the new S2 as bf16 hi/lo image rows d = 16w + r (read after the next B1) ----
        if (c + 1 < c_end) publish_s2();
    };

    f32x4 aq[4], ak[4], av[4];
    if constexpr (PF == 1) {
        issue_loads(aq, ak, av, c_begin * C);
        __syncthreads();
        for (int c = c_begin; c < c_end; ++c) chunk_body(aq, ak, av, c);
    } else {
        f32x4 bq[4], bk[4], bv[4];
        issue_loads(aq, ak, av, c_begin * C);
        if (c_begin + 1 < c_end) issue_loads(bq, bk, bv, (c_begin + 1) * C);
        __syncthreads();
        for (int c = c_begin; c < c_end; c += 2) {
            chunk_body(aq, ak, av, c);
            if (c + 1 < c_end) chunk_body(bq, bk, bv, c + 1);
        }
    }
}

bool mfma_p1_supported(const fastmax_problem& p) {
    return p.p == 1 && p.causal && p.D == 64 && p.in_dtype == FASTMAX_F32 && p.out_dtype == FASTMAX_F32;
}
size_t mfma_p1_workspace(const fastmax_problem& p) { return split_workspace_bytes(p, 64); }

template <int PF, int ST, int SCHED>
static int launch_variant(const MfmaParams& prm, int nblocks, hipStream_t stream) {
    static bool attr_set = false;
    auto kern = fwd_p1_mfma_d64_f32_kernel<PF, ST, SCHED>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           m64::LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(256), m64::LDS_BYTES, stream, prm);
    return (int)hipGetLastError();
}

int launch_fwd_mfma_p1(const FwdArgs& a) {
    if (!mfma_p1_supported(a.prob)) return FASTMAX_E_BAD_SHAPE;
    // tuning key "mfma_variant" (FASTMAX_MFMA_VARIANT at load): 200 = second-generation kernel (fastmax_mfma_v2.hip), 209 its
    // memory-only ablation; 1xx / 2xx = this file's <prefetch distance><staged stores><schedule> variants, for A/B runs
    const int variant = tune_get(TUNE_MFMA_VARIANT);
    if (variant >= 200 && variant <= 209 && mfma_p1_v2_supported(a))
        return launch_fwd_mfma_p1_v2(a, variant == 209 ? 1 : variant - 200);      // 201 / 202 / 203: A/B forms, see there
    const SplitPlan plan = split_plan(a.prob);
    if (plan.nseg > 1) {
        if (!a.workspace || a.workspace_bytes < split_workspace_bytes(a.prob, 64)) return FASTMAX_E_WORKSPACE;
        const int rc = launch_split_states(a, plan, 64, nullptr);
        if (rc) return rc;
    }
    MfmaParams prm{a.q, a.k, a.v, a.qs, a.ks, a.vs, reinterpret_cast<float*>(a.o), a.g, a.prob.H, a.prob.Nq, a.prob.a,
                   reinterpret_cast<const float*>(a.workspace), plan.nseg, plan.cps};
    const int nb = a.prob.B * a.prob.H * plan.nseg;
    switch (variant) {
        case 110: return launch_variant<1, 1, 0>(prm, nb, a.stream);     // <prefetch><staged stores><batched schedule>
        case 210: return launch_variant<2, 1, 0>(prm, nb, a.stream);
#ifdef FASTMAX_ABLATIONS
        case 129: return launch_variant<1, 2, 9>(prm, nb, a.stream);     // memory-pattern ablations: timing only, WRONG RESULTS
        case 119: return launch_variant<1, 1, 9>(prm, nb, a.stream);
        case 219: return launch_variant<2, 1, 9>(prm, nb, a.stream);
#endif
        case 111: return launch_variant<1, 1, 1>(prm, nb, a.stream);
        default: return launch_variant<1, 2, 1>(prm, nb, a.stream);      // 121: batched schedule, staged non-temporal stores
    }
}

}  // namespace fastmax
