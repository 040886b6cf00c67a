// fastmax p=1 masked forward, D = 64, fp32 I/O -- second generation of the headline kernel (gfx950).
//
// Same algorithm and work split as fastmax_mfma.hip (64-token chunks, carried S2 = sum k v^T in MFMA accumulators, split-bf16
// products, attention_mechanisms/fastmax.py:236-241 and 306-312 without the (N,D,D) temporaries), rebuilt to ISSUE LESS per
// chunk.  Measured reason (profiles/r02_transient.md): from an idle device the chip runs the first launches at boost clock,
// then its power controller pulls the clock down and lets it recover over ~60 launches -- but only when the kernel does
// matrix / vector work; the same memory traffic with no arithmetic shows no dip.  Inside the dip the kernel is issue-bound,
// not HBM-bound, so every vector / LDS instruction removed counts twice: fewer cycles at the low clock and a shallower dip.
//
// What changed against the first generation:
//   * Q is never staged: a wave's 16 query rows go from global memory straight into split B fragments (16 floats per
//     lane).  To keep every LDS fragment read a single ds_read_b128, the contraction index m is carried in a PERMUTED
//     order: position p = 32 blk + 8 q4 + e holds m = 32 blk + 4 q4 + e (e < 4) or 32 blk + 16 + 4 q4 + (e - 4): a lane's
//     eight k-values of a 32-wide MFMA step are then two 16-byte pieces of its global Q row, and four lanes cover 64
//     contiguous bytes.  The K image stores its columns, and the S2^T image therefore its columns too, in that order
//     (any order works as long as both operands of a contraction agree).
//   * the result tile still leaves by whole rows, through a wave-private 4 KB of LDS (stores straight from the
//     accumulator layout -- 64-byte pieces of 16 rows per instruction -- cost 7 % of the launch, measured).
//   * global addresses ride on buffer descriptors (wave-uniform base + per-lane 32-bit offset + scalar chunk offset):
//     the 64-bit multiply / add chains per load are gone from the vector ALU.
//   * the causal mask touches only the diagonal 16x16 tile (scalar branches on the wave's tile index), not all four.
//   * q'.ksum is 16 fused multiply-adds on the fragment floats, folded into the two shuffles the score sum needs anyway
//     (was 16 DPP adds + LDS writes + a read).
#include "fastmax_mfma_common.h"

namespace fastmax {

struct MfmaV2Params {
    const void *q, *k, *v;
    Strides3 qs, ks, vs;
    float* o;
    float* g;
    int H, N;
    float a;
    const float* state;   // sequence split: inclusive prefix states [(bh*(nseg-1) + seg-1)][64*64 + 64 + 64], or null
    int nseg, cps;        // segments per head, chunks per segment
};

namespace m64v2 {
constexpr int D = 64, C = 64, NT = 256;
constexpr int IMG = C * D * 2;               // one bf16 image: 8 KiB
constexpr int KH = 0, KL = IMG, VH = 2 * IMG, VL = 3 * IMG;
constexpr int S2H = 4 * IMG, S2L = 5 * IMG;  // [d][position of m] images of S2^T
constexpr int S1V = 6 * IMG;                 // 2 x 64 floats (double-buffered by chunk parity)
constexpr int KSUM = S1V + 512;              // 2 x 64 floats
constexpr int PARTV = KSUM + 512;            // 16 x 64 floats: per-row-group column sums of V
constexpr int PARTK = PARTV + 4096;
constexpr int OST = PARTK + 4096;            // 4 x 4 KiB: per-wave staging of the result tile
constexpr int LDS_BYTES = OST + 16384;       // 74752
}  // namespace m64v2

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// x0, x1 -> packed bf16 (hi0, hi1) rounded to nearest and packed bf16 of the residuals: 6 vector instructions
// (the pack is opaque to the compiler on purpose: left to itself it converts x0 a second time to avoid the shift's dependency)
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned int& hi, unsigned int& lo) {
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hi) : "v"(x0), "v"(x1));
    const float f0 = __uint_as_float(hi << 16), f1 = __uint_as_float(hi & 0xffff0000u);
    const bf16x2 l = {(__bf16)(x0 - f0), (__bf16)(x1 - f1)};
    lo = __builtin_bit_cast(unsigned int, l);
}
__device__ __forceinline__ void split_quad(const f32x4 x, u32x2& hi, u32x2& lo) {
    unsigned int h0, l0, h1, l1;
    split_pair(x[0], x[1], h0, l0);
    split_pair(x[2], x[3], h1, l1);
    hi = u32x2{h0, h1};
    lo = u32x2{l0, l1};
}
// NMF = 3: the three-term split product.  NMF = 1 / 0 (timing-only ablations, wrong results): hi.hi only / no matrix
// instruction at all (operands kept live), to see which unit's power produces the clock dip after an idle start
template <int NMF>
__device__ __forceinline__ f32x4 prod(const bf16x8 ah, const bf16x8 al, const bf16x8 bh, const bf16x8 bl, f32x4 c) {
    if constexpr (NMF == 3) return mfma3(ah, al, bh, bl, c);
    else if constexpr (NMF == 1) {
        asm volatile("" ::"v"(al), "v"(bl));
        return mfma(ah, bh, c);
    } else {
        asm volatile("" ::"v"(ah), "v"(al), "v"(bh), "v"(bl));
        return c;
    }
}
__device__ __forceinline__ bf16x8 pack8(const u32x2 a, const u32x2 b) {
    const u32x4 v = {a[0], a[1], b[0], b[1]};
    return __builtin_bit_cast(bf16x8, v);
}

// ------------------------------------------------------------------------------------------------
// grid = B*H*nseg workgroups, block = 256 threads, dynamic LDS = m64v2::LDS_BYTES.  float32 I/O, D = 64.
// ABL: 0 = the kernel; 1 = timing-only ablation (same loads / stores, no LDS / MFMA work; wrong results)
// RAGGED: false = N is a multiple of 64 (no bounds code anywhere in the loop); true = the last chunk may run past N
// ------------------------------------------------------------------------------------------------
template <int ABL, bool RAGGED, int NMF = 3>
__global__ __launch_bounds__(256, 2) void fwd_p1_d64_f32_v2_kernel(MfmaV2Params prm) {
    using namespace m64v2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave id, provably uniform
    const int r = lane & 15, q4 = lane >> 4;
    const int bh = blockIdx.x / prm.nseg, seg = blockIdx.x - bh * prm.nseg;
    const int b = bh / prm.H, h = bh % prm.H;
    const int N = prm.N;
    const float a = prm.a;

    // buffer descriptors: one per tensor of this head (wave-uniform); offsets below are bytes, 32-bit
    const float* qb = reinterpret_cast<const float*>(prm.q) + (int64_t)b * prm.qs.sb + (int64_t)h * prm.qs.sh;
    const float* kb = reinterpret_cast<const float*>(prm.k) + (int64_t)b * prm.ks.sb + (int64_t)h * prm.ks.sh;
    const float* vb = reinterpret_cast<const float*>(prm.v) + (int64_t)b * prm.vs.sb + (int64_t)h * prm.vs.sh;
    float* ob = prm.o + (int64_t)bh * N * D;
    float* gb = prm.g ? prm.g + (int64_t)bh * N : nullptr;
    const int qsn = (int)prm.qs.sn * 4, ksn = (int)prm.ks.sn * 4, vsn = (int)prm.vs.sn * 4;   // row strides in bytes
    const __amdgpu_buffer_rsrc_t qd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(qb), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t kd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(kb), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t vd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(vb), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t od = __builtin_amdgcn_make_buffer_rsrc(ob, 0, 0x7fffffff, 0x00020000);

    // staging map of K, V: thread -> (row srow + 16u, float4 column scol); a wave-instruction covers 4 whole rows
    const int srow = tid >> 4, scol = tid & 15;
    const int kvoff = srow * ksn + 16 * scol, vvoff = srow * vsn + 16 * scol;
    // fragment map of Q: lane (r, q4) -> row r of the wave's query tile, floats 32 ks + 16 hf + 4 q4 .. + 3
    const int qvoff = r * qsn + 16 * q4;
    const int ovoff = r * (D * 4) + 16 * q4;                     // result: row r, floats 16 dt + 4 q4 .. + 3

    auto issue_kv = [&](u32x4 (&rk)[4], u32x4 (&rv)[4], int n0) {
        if (!RAGGED || n0 + C <= N) {                                // full chunk (block-uniform)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                rk[u] = __builtin_amdgcn_raw_buffer_load_b128(kd, kvoff, (n0 + 16 * u) * ksn, 2);   // aux 2: non-temporal
                rv[u] = __builtin_amdgcn_raw_buffer_load_b128(vd, vvoff, (n0 + 16 * u) * vsn, 2);
            }
        } else {                                                     // last, ragged chunk: rows >= N read row N-1, zeroed
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int gn = n0 + srow + 16 * u;
                const int gc = gn < N ? gn : N - 1;
                rk[u] = __builtin_amdgcn_raw_buffer_load_b128(kd, gc * ksn + 16 * scol, 0, 2);
                rv[u] = __builtin_amdgcn_raw_buffer_load_b128(vd, gc * vsn + 16 * scol, 0, 2);
                if (gn >= N) { rk[u] = u32x4{0, 0, 0, 0}; rv[u] = u32x4{0, 0, 0, 0}; }
            }
        }
    };
    auto issue_q = [&](u32x4 (&rq)[4], int n0, int tile) {
        const int row0 = n0 + 16 * tile;
        if (!RAGGED || row0 + 16 <= N) {                             // wave-uniform
#pragma unroll
            for (int i = 0; i < 4; ++i) rq[i] = __builtin_amdgcn_raw_buffer_load_b128(qd, qvoff + 64 * i, row0 * qsn, 2);
        } else {
            const int gn = row0 + r;
            const int gc = gn < N ? gn : N - 1;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                rq[i] = __builtin_amdgcn_raw_buffer_load_b128(qd, gc * qsn + 16 * q4 + 64 * i, 0, 2);
                if (gn >= N) rq[i] = u32x4{0, 0, 0, 0};
            }
        }
    };

    const int nchunks = (N + C - 1) / C;
    const int c_begin = seg * prm.cps, c_end = min(nchunks, c_begin + prm.cps);
    f32x4 s2acc[4];                                                  // S2[position 16mt + 4q4 + reg][d = 16w + r]
    auto publish_s2 = [&]() {                                        // accumulators -> bf16 hi/lo image rows d = 16w + r
        // the four swizzled offsets are rebuilt from one register per call (2 instructions each): kept as loop invariants
        // they are the four registers that tip the allocation into scratch
        int key = (q4 >> 1) ^ (r & 7);
        asm volatile("" : "+v"(key));
        const int rowb = (16 * w + r) * 128 + ((q4 & 1) << 3);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            u32x2 hi, lo;
            split_quad(s2acc[mt], hi, lo);
            const int off = rowb + (((key ^ (2 * mt)) & 7) << 4);
            *reinterpret_cast<u32x2*>(smem + S2H + off) = hi;
            *reinterpret_cast<u32x2*>(smem + S2L + off) = lo;
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
        // sequence split: start from the prefix state of all earlier segments (record rows are m in natural order)
        const float* rec = prm.state + ((int64_t)bh * (prm.nseg - 1) + (seg - 1)) * (64 * 64 + 128);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            // position p = 16mt + 4q4 + i  ->  m = 32 (p >> 5) + (e < 4 ? 4 q' + e : 16 + 4 q' + e - 4),  q' = (p & 31) >> 3, e = p & 7
            const int m0 = 32 * (mt >> 1) + ((q4 & 1) ? 16 : 0) + 4 * (2 * (mt & 1) + (q4 >> 1));
#pragma unroll
            for (int i = 0; i < 4; ++i) s2acc[mt][i] = rec[(m0 + i) * 64 + 16 * w + r];
        }
        publish_s2();
        if (tid < 64) {
            reinterpret_cast<float*>(smem + S1V)[64 * (c_begin & 1) + tid] = rec[64 * 64 + tid];
            reinterpret_cast<float*>(smem + KSUM)[64 * (c_begin & 1) + tid] = rec[64 * 64 + 64 + tid];
        }
    }

    // image offsets of this thread's staged pieces (row srow + 16u adds 2048u bytes: the swizzle key is row & 7)
    const int koff = img_off<64>(srow, ((scol >> 3) << 2) | (scol & 3)) + (((scol >> 2) & 1) << 3);   // permuted columns
    const int voff = img_off<64>(srow, scol >> 1) + ((scol & 1) << 3);

    u32x4 rq[4], rk[4], rv[4];
    issue_q(rq, c_begin * C, (w + c_begin) & 3);
    issue_kv(rk, rv, c_begin * C);
    __syncthreads();

    for (int c = c_begin; c < c_end; ++c) {
        const int n0 = c * C;
        const int wq = (w + c) & 3;                                  // query tile of this wave in this chunk (rotates)
        if constexpr (ABL != 0) {
            // ABLATION (timing only, wrong results): HBM access patterns with no LDS / MFMA work.
            // ABL 1: the kernel's own (Q as fragments, result from the accumulator layout); 2: Q by whole rows, result as 1;
            // 3: Q as 1, result by whole rows; 4: both by whole rows (the first generation's pattern)
            constexpr bool QROWS = ABL == 2 || ABL == 4, OROWS = ABL == 3 || ABL == 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4 s = __builtin_bit_cast(f32x4, rq[i]) + __builtin_bit_cast(f32x4, rk[i]) + __builtin_bit_cast(f32x4, rv[i]);
                if constexpr (OROWS) {
                    if (!RAGGED || n0 + srow + 16 * i < N)
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, s), od, srow * (D * 4) + 16 * scol, (n0 + 16 * i) * (D * 4), 2);
                } else {
                    if (!RAGGED || n0 + 16 * wq + r < N)
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, s), od, ovoff + 64 * i, (n0 + 16 * wq) * (D * 4), 2);
                }
            }
            if (c + 1 < c_end) {
                if constexpr (QROWS) {
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        rq[u] = __builtin_amdgcn_raw_buffer_load_b128(qd, srow * qsn + 16 * scol, (n0 + C + 16 * u) * qsn, 2);
                } else {
                    issue_q(rq, n0 + C, (w + c + 1) & 3);
                }
                issue_kv(rk, rv, n0 + C);
            }
            continue;
        }
        const int cur = c & 1, nxt = cur ^ 1;
        const float* ksum_cur = reinterpret_cast<const float*>(smem + KSUM) + 64 * cur;
        const float* s1v_cur = reinterpret_cast<const float*>(smem + S1V) + 64 * cur;

        // ---- (a) Q registers -> split B fragments; q'.ksum_prev partial (this lane's 16 columns) -------------
        bf16x8 qh[2], ql[2];
        float gsum = 0.f;
        {
            u32x2 h[4], l[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4 xq = __builtin_bit_cast(f32x4, rq[i]) * a;
                const f32x4 ks4 = *reinterpret_cast<const f32x4*>(ksum_cur + 16 * i + 4 * q4);
                gsum = fmaf(xq[0], ks4[0], gsum);
                gsum = fmaf(xq[1], ks4[1], gsum);
                gsum = fmaf(xq[2], ks4[2], gsum);
                gsum = fmaf(xq[3], ks4[3], gsum);
                split_quad(xq, h[i], l[i]);
            }
            qh[0] = pack8(h[0], h[1]); ql[0] = pack8(l[0], l[1]);
            qh[1] = pack8(h[2], h[3]); ql[1] = pack8(l[2], l[3]);
        }
        __builtin_amdgcn_sched_barrier(0);                           // Q is consumed BEFORE the refills below are issued

        // ---- (b) K, V registers -> bf16 hi/lo images; exact fp32 column sums -------------------------------
        {
            f32x4 ck = {0, 0, 0, 0}, cv = {0, 0, 0, 0};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const f32x4 xk = __builtin_bit_cast(f32x4, rk[u]), xv = __builtin_bit_cast(f32x4, rv[u]);
                u32x2 hi, lo;
                split_quad(xk, hi, lo);
                *reinterpret_cast<u32x2*>(smem + KH + koff + 2048 * u) = hi;
                *reinterpret_cast<u32x2*>(smem + KL + koff + 2048 * u) = lo;
                split_quad(xv, hi, lo);
                *reinterpret_cast<u32x2*>(smem + VH + voff + 2048 * u) = hi;
                *reinterpret_cast<u32x2*>(smem + VL + voff + 2048 * u) = lo;
                ck += xk;
                cv += xv;
            }
            *reinterpret_cast<f32x4*>(smem + PARTK + (srow * 64 + 4 * scol) * 4) = ck;
            *reinterpret_cast<f32x4*>(smem + PARTV + (srow * 64 + 4 * scol) * 4) = cv;
        }
        __builtin_amdgcn_sched_barrier(0);
        // refill both register sets one chunk ahead, Q first (it is consumed first next time).  Q used to be requested right after
        // its consumption, i.e. BEFORE the K / V registers were staged: the memory counter is in-order, so the wait for the last
        // K / V piece (vmcnt(0)) also waited for the four Q loads issued a moment earlier -- their latency exposed in every chunk
        if (c + 1 < c_end) {
            issue_q(rq, n0 + C, (w + c + 1) & 3);
            issue_kv(rk, rv, n0 + C);
        }
        __builtin_amdgcn_sched_barrier(0);
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

        // ---- phase A: 16 queries per wave.  Plain loops: the compiler's own schedule (the hand-pinned batches of the first
        // generation measured the same and need more registers than two waves per SIMD leave) -----------------------
        f32x4 oacc[4];
        bf16x8 ph[2], pl[2];
        // (3) inter-chunk: O^T = S1 + S2^T Q'^T      (A = S2^T image rows d, B = Q'^T fragments)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            oacc[dt] = *reinterpret_cast<const f32x4*>(s1v_cur + 16 * dt + 4 * q4);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16x8 sh = ld_row8<64>(smem, S2H, 16 * dt + r, 4 * ks + q4);
                const bf16x8 sl = ld_row8<64>(smem, S2L, 16 * dt + r, 4 * ks + q4);
                oacc[dt] = prod<NMF>(sh, sl, qh[ks], ql[ks], oacc[dt]);
            }
        }
        // (1) scores S^T[j][i] = k_j . q'_i for key tiles jt <= wq; P = 1 + s below the diagonal tile, masked on it, 0 above
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            f32x4 pt[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int jt = 2 * s + e;
                if (jt <= wq) {                                      // wave-uniform
                    f32x4 sc = {0, 0, 0, 0};
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        const bf16x8 kh = ld_row8<64>(smem, KH, 16 * jt + r, 4 * ks + q4);
                        const bf16x8 kl = ld_row8<64>(smem, KL, 16 * jt + r, 4 * ks + q4);
                        sc = prod<NMF>(kh, kl, qh[ks], ql[ks], sc);
                    }
                    if (jt < wq) {
                        gsum += (sc[0] + sc[1]) + (sc[2] + sc[3]);
                        pt[e] = sc + 1.0f;
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const bool keep = (4 * q4 + i) <= r;
                            const float sv = keep ? sc[i] : 0.f;
                            gsum += sv;
                            pt[e][i] = keep ? 1.0f + sv : 0.f;
                        }
                    }
                } else {
                    pt[e] = f32x4{0, 0, 0, 0};
                }
            }
            u32x2 h0, l0, h1, l1;
            split_quad(pt[0], h0, l0);
            split_quad(pt[1], h1, l1);
            ph[s] = pack8(h0, h1);
            pl[s] = pack8(l0, l1);
        }
        // (2) intra-chunk: O^T += V^T P^T   (A = V^T by transposed reads, B = P^T from registers)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (2 * s <= wq) {                                       // wave-uniform
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const bf16x8 vh = ld_tr8<64>(smem, VH, 32 * s, 16 * dt, lane);
                    const bf16x8 vl = ld_tr8<64>(smem, VL, 32 * s, 16 * dt, lane);
                    oacc[dt] = prod<NMF>(vh, vl, ph[s], pl[s], oacc[dt]);
                }
            }
        }
        // denominator: count + q'.ksum_prev + intra-chunk score sum (over the 4 k-groups of the lane's query)
        gsum += __shfl_xor(gsum, 16, 64);
        gsum += __shfl_xor(gsum, 32, 64);
        const int qi = 16 * wq + r, gi = n0 + qi;
        const float gval = (float)(gi + 1) + gsum;
        const float ginv = 1.0f / gval;
        // Result tile -> global memory by WHOLE ROWS: the 16 x 64 tile goes through this wave's private 4 KB of LDS (no
        // barrier: only this wave touches it) so that every store instruction writes four complete 256-byte rows.  Stores
        // from the accumulator layout (64-byte pieces of 16 rows per instruction) measured 412 us against 385 us for the
        // memory-only ablation of this kernel (profiles/r02_headline_v2.md).  Plain global stores, not buffer stores: a
        // buffer_store_dwordx4 with an SGPR offset followed directly by a vector write of its data registers stored
        // corrupted data on gfx950 (the compiler's hazard recogniser exempts that form).
        {
            char* ost = smem + OST + 4096 * w;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
                *reinterpret_cast<f32x4*>(ost + r * 256 + ((((4 * dt + q4) ^ r) & 15) << 4)) = oacc[dt] * ginv;
            float* orow = ob + (int64_t)(n0 + 16 * wq) * D;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int rl = 4 * u + q4, c16 = r;                  // row inside the wave tile, 16-byte column
                const f32x4 val = *reinterpret_cast<const f32x4*>(ost + rl * 256 + (((c16 ^ rl) & 15) << 4));
                if (!RAGGED || n0 + 16 * wq + rl < N)
                    __builtin_nontemporal_store(val, reinterpret_cast<f32x4*>(orow + rl * D + 4 * c16));
            }
            if ((!RAGGED || gi < N) && gb && q4 == 0) gb[gi] = gval;
        }

        // ---- phase B: S2[:, 16w..16w+15] += K^T V  (A = K^T, B = V, both by transposed reads) --------
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const bf16x8 vh = ld_tr8<64>(smem, VH, 32 * s, 16 * w, lane);
            const bf16x8 vl = ld_tr8<64>(smem, VL, 32 * s, 16 * w, lane);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const bf16x8 kh = ld_tr8<64>(smem, KH, 32 * s, 16 * mt, lane);
                const bf16x8 kl = ld_tr8<64>(smem, KL, 32 * s, 16 * mt, lane);
                s2acc[mt] = prod<NMF>(kh, kl, vh, vl, s2acc[mt]);
            }
        }
        __syncthreads();                                             // B2: every read of this chunk's images is done

        // ---- (e) publish the new S2 as bf16 hi/lo image rows d = 16w + r (read after the next B1) ----
        if (c + 1 < c_end) publish_s2();
    }
}

bool mfma_p1_v2_supported(const FwdArgs& a) {
    const fastmax_problem& p = a.prob;
    if (!(p.p == 1 && p.causal && p.D == 64 && p.in_dtype == FASTMAX_F32 && p.out_dtype == FASTMAX_F32)) return false;
    // 32-bit byte offsets inside one head
    const int64_t lim = (int64_t)1 << 31;
    return (int64_t)p.Nq * a.qs.sn * 4 < lim && (int64_t)p.Nq * a.ks.sn * 4 < lim && (int64_t)p.Nq * a.vs.sn * 4 < lim;
}

template <int ABL, bool RAGGED, int NMF = 3>
static int launch_v2_variant(const MfmaV2Params& prm, int nblocks, hipStream_t stream) {
    static bool attr_set = false;
    auto kern = fwd_p1_d64_f32_v2_kernel<ABL, RAGGED, NMF>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           m64v2::LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(256), m64v2::LDS_BYTES, stream, prm);
    return (int)hipGetLastError();
}

int launch_fwd_mfma_p1_v2(const FwdArgs& a, int ablation) {
    if (!mfma_p1_v2_supported(a)) return FASTMAX_E_BAD_SHAPE;
    const SplitPlan plan = split_plan(a.prob);
    if (plan.nseg > 1) {
        if (!a.workspace || a.workspace_bytes < split_workspace_bytes(a.prob, 64)) return FASTMAX_E_WORKSPACE;
        const int rc = launch_split_states(a, plan, 64, nullptr);
        if (rc) return rc;
    }
    MfmaV2Params prm{a.q, a.k, a.v, a.qs, a.ks, a.vs, reinterpret_cast<float*>(a.o), a.g, a.prob.H, a.prob.Nq, a.prob.a,
                     reinterpret_cast<const float*>(a.workspace), plan.nseg, plan.cps};
    const int nb = a.prob.B * a.prob.H * plan.nseg;
    const bool ragged = (a.prob.Nq & 63) != 0;
#ifdef FASTMAX_ABLATIONS
    switch (ablation) {                                              // timing-only ablations: WRONG RESULTS, ablation builds only
        case 1: return launch_v2_variant<1, true>(prm, nb, a.stream);        // memory passes only
        case 6: return launch_v2_variant<2, true>(prm, nb, a.stream);
        case 7: return launch_v2_variant<3, true>(prm, nb, a.stream);
        case 8: return launch_v2_variant<4, true>(prm, nb, a.stream);
        case 4: return launch_v2_variant<0, true, 1>(prm, nb, a.stream);      // hi.hi products only
        case 5: return launch_v2_variant<0, true, 0>(prm, nb, a.stream);      // no matrix instructions
        default: break;
    }
#else
    (void)ablation;
#endif
    return ragged ? launch_v2_variant<0, true>(prm, nb, a.stream) : launch_v2_variant<0, false>(prm, nb, a.stream);
}

}  // namespace fastmax
