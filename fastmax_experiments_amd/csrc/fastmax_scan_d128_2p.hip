// fastmax p=1 masked BACKWARD in linear time for fp32 / fp16 tensors at 64 < D <= 128 (two bf16 parts per operand).
//
// The hand-derived gradients of attention_mechanisms/fastmax.py:432-485 (dQ), 541-604 (dK), 647-691 (dV), with
// w_i = 1/g_i, c_i = G_i.o_i, ghat_i = w_i G_i, e_i = -w_i c_i, are three instances of ONE causal scan
//     F_r = sum_{c in range(r)} (alpha + beta x_r.y_c) z_c
//   dQ_i = a sum_{j<=i} (e_i + ghat_i.v_j) k_j       forwards : x = ghat, y = v,    z = k,    alpha = e of the OUTPUT row
//   dV_j =   sum_{i>=j} (1 + a k_j.q_i) ghat_i       backwards: x = k,    y = q,    z = ghat, alpha = 1
//   dK_j = a sum_{i>=j} (e_i + v_j.ghat_i) q_i       backwards: x = v,    y = ghat, z = q,    alpha = e of the SUMMED row
// -- the shape of the forward itself (x = q, y = k, z = v, alpha = 1, then a division by g).  So the backward at this head
// size is the forward kernel of fastmax_mfma_d128_2p.hip three times with different operand roles: 64-token chunks, carried
// state S2 = sum y z^T (128 x 128, fp32 in MFMA accumulators, bf16 hi/lo image in LDS), S1 = sum alpha_c z_c, a masked
// 64 x 64 tile inside the chunk, x rows straight from global memory into B fragments, no (N,D,D) temporaries.  The fused
// two-kernel form used at D <= 64 (fastmax_mfma_bwd_lin.hip) needs four hi/lo images + the state = more than 160 KB here.
// Cost: 9 tensor reads + 3 writes instead of 5 + 3, against the O(N^2) tiles these shapes fell back to before.
#include "fastmax_mfma_common.h"

namespace fastmax {

enum ScanMode { SCAN_DQ = 0, SCAN_DV = 1, SCAN_DK = 2 };

struct ScanParams {
    const void *x, *y, *z;          // x: row operand of the output rows (fragments from global memory); y, z: staged
    Strides3 xs, ys, zs;
    void* out;                      // (B,H,N,D) contiguous, dtype of the inputs
    const float *wbuf, *ebuf;       // w_i = 1/g_i, e_i = -w_i c_i   (B,H,N)
    int H, N, D, dtype;
    float beta, out_scale;
    // sequence split (few heads): workgroup = (head, segment of `cps` chunks).  A first launch (STATE_ONLY) leaves every
    // segment's own sums in `state`; the main launch starts segment s from the sum of the records before it (after it, for
    // the reverse scans).  Record: S2 as the accumulators hold it ([mt][wave][lane] x 4 floats) + S1 (128 floats).
    float* state;
    int nseg, cps;
};
constexpr int kScanRec = 8 * 8 * 64 * 4 + 128;       // floats per (head, segment) record

// w, e of every row: one wave per row.  grid = (ceil(N/4), B*H), block = 256
template <typename TIN>
__global__ __launch_bounds__(256) void scan_prep_kernel(const void* go, Strides3 gos, const void* o, const float* g, float* wbuf,
                                                        float* ebuf, int H, int N, int D) {
    constexpr int EPL = InTraits<TIN>::EPL;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int bh = blockIdx.y, b = bh / H, h = bh % H;
    const int i = blockIdx.x * 4 + wave;
    if (i >= N) return;
    const TIN* grow = reinterpret_cast<const TIN*>(go) + (int64_t)b * gos.sb + (int64_t)h * gos.sh + (int64_t)i * gos.sn;
    const TIN* orow = reinterpret_cast<const TIN*>(o) + ((int64_t)bh * N + i) * D;
    float s = 0.f;
    if (lane * EPL < D) {
        float x[EPL], y[EPL];
        piece_to_float<TIN>(*reinterpret_cast<const u32x4*>(grow + lane * EPL), x);
        piece_to_float<TIN>(*reinterpret_cast<const u32x4*>(orow + lane * EPL), y);
#pragma unroll
        for (int e = 0; e < EPL; ++e) s = fmaf(x[e], y[e], s);
    }
    s = wave_sum(s);
    if (lane == 0) {
        const float w = 1.0f / g[(int64_t)bh * N + i];
        wbuf[(int64_t)bh * N + i] = w;
        ebuf[(int64_t)bh * N + i] = -w * s;
    }
}

template <typename TIN, int MODE, bool STATE_ONLY, bool BUF>
__global__ __launch_bounds__(512, 1) void scan_d128_2p_kernel(ScanParams prm) {
    constexpr int NP = 2, EPL = InTraits<TIN>::EPL;
    static_assert(InTraits<TIN>::NP == 2, "two-part operands");
    constexpr bool REV = MODE != SCAN_DQ;
    constexpr bool XSCALE = MODE == SCAN_DQ, YSCALE = MODE == SCAN_DK, ZSCALE = MODE == SCAN_DV;
    constexpr int DP = 128, C = 64, IMG = C * DP * 2, SIMG = DP * DP * 2;
    constexpr int YI = 0, ZI = NP * IMG, S2I = 2 * NP * IMG, S1V = S2I + 2 * SIMG, EV = S1V + DP * 4;    // EV: e of the chunk's 64 rows
    constexpr int COLS = DP / EPL, RPP = 512 / COLS, NPASS = C / RPP;
    constexpr int KS = DP / 32, MT = DP / 16, QL = 8 / EPL;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qt = w & 3, dh = w >> 2;
    const int r = lane & 15, q4 = lane >> 4;
    const int bh = blockIdx.x / prm.nseg, seg = blockIdx.x - bh * prm.nseg, b = bh / prm.H, h = bh % prm.H;
    const int N = prm.N, D = prm.D;
    if constexpr (STATE_ONLY) {
        if (seg == (REV ? 0 : prm.nseg - 1)) return;                 // nobody starts from this segment's sums
    }
    const TIN* xb = reinterpret_cast<const TIN*>(prm.x) + (int64_t)b * prm.xs.sb + (int64_t)h * prm.xs.sh;
    const TIN* yb = reinterpret_cast<const TIN*>(prm.y) + (int64_t)b * prm.ys.sb + (int64_t)h * prm.ys.sh;
    const TIN* zb = reinterpret_cast<const TIN*>(prm.z) + (int64_t)b * prm.zs.sb + (int64_t)h * prm.zs.sh;
    const float* wb = prm.wbuf + (int64_t)bh * N;
    const float* eb = prm.ebuf + (int64_t)bh * N;
    const int srow = tid / COLS, scol = tid % COLS;

    u32x4 ry[NPASS], rz[NPASS], rx[KS][QL];
    float rws[NPASS];               // w of the staged rows (y or z scaled by it)
    float rwx = 0.f, rex = 0.f;     // w, e of this lane's output row
    float rec = 0.f;                // e of row tid of the chunk (threads 0..63), for EV
    const ScanLoader<BUF, TIN, NPASS, RPP> yload(yb, prm.ys.sn, N, D, DP, srow, scol), zload(zb, prm.zs.sn, N, D, DP, srow, scol);
    const RowPieceLoader<BUF, TIN> xrows(xb, prm.xs.sn, N, D);
    auto issue = [&](int c) {
        yload.load(c, ry);
        zload.load(c, rz);
        const int row = c * C + 16 * qt + r;                       // this wave's output row on this lane
        if constexpr (!STATE_ONLY) {
            int q4o = q4;                                          // opaque: piece offsets re-formed per chunk (see fastmax_mfma_d128_2p.hip)
            asm volatile("" : "+v"(q4o));
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int u = 0; u < QL; ++u) rx[ks][u] = xrows.load(row, (32 * ks + 8 * q4o) / EPL + u);
        }
        if constexpr (YSCALE || ZSCALE) {
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps) {
                const int sr = c * C + srow + ps * RPP;
                rws[ps] = sr < N ? wb[sr] : 0.f;
            }
        }
        if constexpr (MODE == SCAN_DQ && !STATE_ONLY) {
            rwx = row < N ? wb[row] : 0.f;
            rex = row < N ? eb[row] : 0.f;
        }
        if constexpr (MODE == SCAN_DK) {
            if (tid < 64) rec = (c * C + tid) < N ? eb[c * C + tid] : 0.f;
        }
    };
    bf16x8 ones;
#pragma unroll
    for (int i = 0; i < 8; ++i) ones[i] = (__bf16)(r == 0 ? 1.0f : 0.0f);

    const int nchunks = (N + C - 1) / C;
    f32x4 s2acc[MT];               // S2[y index 16mt + 4q4 + reg][z index 16w + r]
    f32x4 s1acc;                   // S1[16w + r] (row 0)
    const float beta = prm.beta;
    auto publish = [&]() {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            bf16x4 hi, lo;
            split4(s2acc[mt] * beta, hi, lo);
            const int off = img_off<DP>(16 * w + r, 2 * mt + (q4 >> 1)) + ((q4 & 1) << 3);
            *reinterpret_cast<bf16x4*>(smem + S2I + off) = hi;
            *reinterpret_cast<bf16x4*>(smem + S2I + SIMG + off) = lo;
        }
        if (q4 == 0) reinterpret_cast<float*>(smem + S1V)[16 * w + r] = s1acc[0];
    };
    s1acc = f32x4{0, 0, 0, 0};
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) s2acc[mt] = f32x4{0, 0, 0, 0};
    const int c_lo = seg * prm.cps, c_hi = min(nchunks, c_lo + prm.cps), nloc = c_hi - c_lo;
    const int c_first = REV ? c_hi - 1 : c_lo, c_step = REV ? -1 : 1;
    bool has_prefix = false;
    if constexpr (!STATE_ONLY) {
        // the sums of the segments this one continues: earlier ones (forward scan) or later ones (reverse scans)
        const int s_lo = REV ? seg + 1 : 0, s_hi = REV ? prm.nseg : seg;
        has_prefix = s_hi > s_lo;
        for (int sg = s_lo; sg < s_hi; ++sg) {
            const float* rec = prm.state + ((int64_t)bh * prm.nseg + sg) * kScanRec;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) s2acc[mt] += *reinterpret_cast<const f32x4*>(rec + ((mt * 8 + w) * 64 + lane) * 4);
            if (q4 == 0) s1acc[0] += rec[8 * 8 * 64 * 4 + 16 * w + r];
        }
    }
    if (has_prefix) {
        publish();
    } else {
        for (int i = tid; i < (2 * SIMG) / 16; i += 512) *reinterpret_cast<f32x4*>(smem + S2I + 16 * i) = f32x4{0, 0, 0, 0};
        if (tid < DP) reinterpret_cast<float*>(smem + S1V)[tid] = 0.f;
    }
    issue(c_first);
    __syncthreads();

    for (int it = 0, c = c_first; it < nloc; ++it, c += c_step) {
        const int n0 = c * C;
        // ---- staging: y and z rows (one of them scaled by w of its row) as hi / lo images; x fragments in registers -------
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int row = srow + ps * RPP;
            if constexpr (YSCALE) {
                float xv[EPL];
                piece_to_float<TIN>(ry[ps], xv);
#pragma unroll
                for (int e = 0; e < EPL; ++e) xv[e] *= rws[ps];
                stage_floats<DP, EPL, NP>(smem, YI, row, scol, xv);
            } else {
                stage_piece<DP, TIN>(smem, YI, row, scol, ry[ps]);
            }
            if constexpr (ZSCALE) {
                float xv[EPL];
                piece_to_float<TIN>(rz[ps], xv);
#pragma unroll
                for (int e = 0; e < EPL; ++e) xv[e] *= rws[ps];
                stage_floats<DP, EPL, NP>(smem, ZI, row, scol, xv);
            } else {
                stage_piece<DP, TIN>(smem, ZI, row, scol, rz[ps]);
            }
        }
        if constexpr (MODE == SCAN_DK) {
            if (tid < 64) reinterpret_cast<float*>(smem + EV)[tid] = rec;
        }
        Frag<NP> xf[KS];
        const float alpha_r = MODE == SCAN_DQ ? rex : 1.0f;          // alpha of this lane's output row (DQ), else unused / 1
        if constexpr (!STATE_ONLY) {
            int q4o = q4;                                          // opaque: piece offsets re-formed per chunk (see fastmax_mfma_d128_2p.hip)
            asm volatile("" : "+v"(q4o));
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                float xq[8];
                if constexpr (QL == 2) {
                    float lo4[4], hi4[4];
                    piece_to_float<TIN>(rx[ks][0], lo4);
                    piece_to_float<TIN>(rx[ks][1], hi4);
#pragma unroll
                    for (int i = 0; i < 4; ++i) { xq[i] = lo4[i]; xq[4 + i] = hi4[i]; }
                } else {
                    piece_to_float<TIN>(rx[ks][0], xq);
                }
                if constexpr (XSCALE) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) xq[i] *= rwx;
                }
                bf16x4 h0, l0, h1, l1;
                split4(f32x4{xq[0], xq[1], xq[2], xq[3]}, h0, l0);
                split4(f32x4{xq[4], xq[5], xq[6], xq[7]}, h1, l1);
                xf[ks].p[0] = cat4(h0, h1);
                xf[ks].p[1] = cat4(l0, l1);
            }
        }
        if (it + 1 < nloc) issue(c + c_step);
        __syncthreads();                                             // B1
        if constexpr (!STATE_ONLY) {
        // ---- phase A: output rows of tile qt, output columns of d-half dh ---------------------------------------------
        f32x4 oacc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int dt = 4 * dh + t;
            // S1 holds sum z_c (DQ, DV) or sum e_c z_c (DK); DQ multiplies by the output row's own e
            f32x4 acc = *reinterpret_cast<const f32x4*>(smem + S1V + (16 * dt + 4 * q4) * 4) * alpha_r;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                Frag<2> sf;
                sf.p[0] = *reinterpret_cast<const bf16x8*>(smem + S2I + img_off<DP>(16 * dt + r, 4 * ks + q4));
                sf.p[1] = *reinterpret_cast<const bf16x8*>(smem + S2I + SIMG + img_off<DP>(16 * dt + r, 4 * ks + q4));
                acc = mfma_parts<2, 2>(sf, xf[ks], acc);
            }
            oacc[t] = acc;
        }
        Frag<2> pf[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            f32x4 pt[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int jt = 2 * s + e;
                const bool live_tile = REV ? (jt >= qt) : (jt <= qt);
                f32x4 sc = {0, 0, 0, 0};
                if (live_tile) {
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        Frag<2> yf;
#pragma unroll
                        for (int p = 0; p < NP; ++p) yf.p[p] = ld_row8<DP>(smem, YI + p * IMG, 16 * jt + r, 4 * ks + q4);
                        sc = mfma_parts<2, 2>(yf, xf[ks], sc);
                    }
                }
                f32x4 al = {alpha_r, alpha_r, alpha_r, alpha_r};
                if constexpr (MODE == SCAN_DK) al = *reinterpret_cast<const f32x4*>(smem + EV + (16 * jt + 4 * q4) * 4);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bool diag = REV ? ((4 * q4 + i) >= r) : ((4 * q4 + i) <= r);
                    const bool keep = (REV ? (jt > qt) : (jt < qt)) || (jt == qt && diag);
                    pt[e][i] = keep ? fmaf(beta, sc[i], al[i]) : 0.f;
                }
            }
            bf16x4 h0, l0, h1, l1;
            split4(pt[0], h0, l0);
            split4(pt[1], h1, l1);
            pf[s].p[0] = cat4(h0, h1);
            pf[s].p[1] = cat4(l0, l1);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const bool live_step = REV ? (2 * s + 1 >= qt) : (2 * s <= qt);
            if (live_step) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    Frag<2> zf;
#pragma unroll
                    for (int p = 0; p < NP; ++p) zf.p[p] = ld_tr8<DP>(smem, ZI + p * IMG, 32 * s, 16 * (4 * dh + t), lane);
                    oacc[t] = mfma_parts<2, 2>(zf, pf[s], oacc[t]);
                }
            }
        }
        const int gi = n0 + 16 * qt + r;
        if (gi < N) {                                                // lane: row gi, columns 64 dh + 16 t + 4 q4 .. + 3
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int col = 64 * dh + 16 * t + 4 * q4;
                if (col < D) store4_any(prm.out, prm.dtype, ((int64_t)bh * N + gi) * D + col, oacc[t] * prm.out_scale);
            }
        }
        }   // !STATE_ONLY
        // ---- phase B: S2[:, 16w ..] += Y^T Z;  S1 += sum_c alpha_c z_c  (ones, or the chunk's e, times Z) --------------
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            Frag<2> zf;
#pragma unroll
            for (int p = 0; p < NP; ++p) zf.p[p] = ld_tr8<DP>(smem, ZI + p * IMG, 32 * s, 16 * w, lane);
            if constexpr (MODE == SCAN_DK) {
                // A operand row 0 = e of the 32 tokens of this k-step in the transposed-read order (tokens 32s + 4q4 + i, then + 16)
                bf16x4 h0 = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f}, l0 = h0, h1 = h0, l1 = h0;
                if (r == 0) {
                    split4(*reinterpret_cast<const f32x4*>(smem + EV + (32 * s + 4 * q4) * 4), h0, l0);
                    split4(*reinterpret_cast<const f32x4*>(smem + EV + (32 * s + 16 + 4 * q4) * 4), h1, l1);
                }
                const bf16x8 eh = cat4(h0, h1), el = cat4(l0, l1);
                s1acc = mfma(eh, zf.p[0], s1acc);
                s1acc = mfma(el, zf.p[0], s1acc);
                s1acc = mfma(eh, zf.p[1], s1acc);
            } else {
                s1acc = mfma(ones, zf.p[0], s1acc);
                s1acc = mfma(ones, zf.p[1], s1acc);
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                Frag<2> yf;
#pragma unroll
                for (int p = 0; p < NP; ++p) yf.p[p] = ld_tr8<DP>(smem, YI + p * IMG, 32 * s, 16 * mt, lane);
                s2acc[mt] = mfma_parts<2, 2>(yf, zf, s2acc[mt]);
            }
        }
        __syncthreads();                                             // B2
        if constexpr (!STATE_ONLY) {
            if (it + 1 < nloc) publish();
        }
    }
    if constexpr (STATE_ONLY) {
        float* rec = prm.state + ((int64_t)bh * prm.nseg + seg) * kScanRec;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) *reinterpret_cast<f32x4*>(rec + ((mt * 8 + w) * 64 + lane) * 4) = s2acc[mt];
        if (q4 == 0) rec[8 * 8 * 64 * 4 + 16 * w + r] = s1acc[0];
    }
}

template <typename TIN, int MODE, bool STATE_ONLY, bool BUF>
static int launch_scan_kb(const ScanParams& prm, int nb, hipStream_t stream) {
    constexpr int DP = 128;
    constexpr int lds = 4 * 64 * DP * 2 + 2 * DP * DP * 2 + DP * 4 + 256;
    static_assert(lds <= 160 * 1024, "LDS budget");
    auto kern = scan_d128_2p_kernel<TIN, MODE, STATE_ONLY, BUF>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(nb), dim3(512), lds, stream, prm);
    return (int)hipGetLastError();
}

template <typename TIN, int MODE, bool STATE_ONLY>
static int launch_scan_k(const ScanParams& prm, int nb, hipStream_t stream) {
    // streamed y / z tiles through buffer descriptors when a (b,h) slab fits their 31-bit offsets
    const bool buf = quad32_span_ok(prm.ys.sn, prm.N, prm.D, (int)sizeof(TIN)) && quad32_span_ok(prm.zs.sn, prm.N, prm.D, (int)sizeof(TIN)) &&
                     quad32_span_ok(prm.xs.sn, prm.N, prm.D, (int)sizeof(TIN));
    return buf ? launch_scan_kb<TIN, MODE, STATE_ONLY, true>(prm, nb, stream) : launch_scan_kb<TIN, MODE, STATE_ONLY, false>(prm, nb, stream);
}

// one scan: with a sequence split, the segments' own sums first
template <typename TIN, int MODE>
static int launch_scan_t(const ScanParams& prm, int nb, hipStream_t stream) {
    if (prm.nseg > 1) {
        const int rc = launch_scan_k<TIN, MODE, true>(prm, nb * prm.nseg, stream);
        if (rc) return rc;
    }
    return launch_scan_k<TIN, MODE, false>(prm, nb * prm.nseg, stream);
}

// few heads: segments of at least 8 chunks (512 tokens) until ~256 workgroups exist (one per CU: 132 KB of LDS each)
static void scan_split_plan(const fastmax_problem& p, int& nseg, int& cps) {
    const int nchunks = (p.Nq + 63) / 64, BH = p.B * p.H;
    static const int target = [] { const char* e = getenv("FASTMAX_SCAN_SPLIT_TARGET"); return e ? atoi(e) : 256; }();
    int want = target / (BH > 0 ? BH : 1);
    if (want > nchunks / 8) want = nchunks / 8;
    if (want < 1) want = 1;
    cps = (nchunks + want - 1) / want;
    nseg = (nchunks + cps - 1) / cps;
}

bool scan_bwd_supported(const fastmax_problem& p) {
    // the shapes fastmax_mfma_d128_2p.hip serves forwards: two-part operands at 64 < D <= 128
    if (!(p.p == 1 && p.causal) || p.D <= 64 || p.D > 128 || p.in_dtype != p.out_dtype || p.Nq < 512) return false;
    if (p.in_dtype == FASTMAX_F32) return (p.D % 4) == 0;
    if (p.in_dtype == FASTMAX_F16) return (p.D % 8) == 0;
    return false;
}
size_t scan_bwd_workspace(const fastmax_problem& p) {
    int nseg, cps;
    scan_split_plan(p, nseg, cps);
    return 2 * sizeof(float) * (size_t)p.B * p.H * p.Nq + 64 + (nseg > 1 ? sizeof(float) * (size_t)p.B * p.H * nseg * kScanRec : 0);
}

template <typename TIN>
static int launch_scan_bwd_t(const BwdArgs& a) {
    const fastmax_problem& p = a.prob;
    const int BH = p.B * p.H;
    float* wbuf = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(a.workspace) + 15) & ~(uintptr_t)15);
    float* ebuf = wbuf + (size_t)BH * p.Nq;
    float* state = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(ebuf + (size_t)BH * p.Nq) + 15) & ~(uintptr_t)15);
    int nseg, cps;
    scan_split_plan(p, nseg, cps);
    hipLaunchKernelGGL((scan_prep_kernel<TIN>), dim3((p.Nq + 3) / 4, BH), dim3(256), 0, a.stream, a.grad_o, a.gos, a.o, a.g, wbuf, ebuf,
                       p.H, p.Nq, p.D);
    const Strides3 os{(int64_t)p.H * p.Nq * p.D, (int64_t)p.Nq * p.D, (int64_t)p.D};
    (void)os;
    // dQ: x = grad_o (scaled by w), y = v, z = k
    ScanParams dq{a.grad_o, a.v, a.k, a.gos, a.vs, a.ks, a.dq, wbuf, ebuf, p.H, p.Nq, p.D, p.in_dtype, 1.0f, p.a, state, nseg, cps};
    int rc = launch_scan_t<TIN, SCAN_DQ>(dq, BH, a.stream);
    if (rc) return rc;
    // dV: x = k, y = q, z = grad_o (scaled by w)
    ScanParams dv{a.k, a.q, a.grad_o, a.ks, a.qs, a.gos, a.dv, wbuf, ebuf, p.H, p.Nq, p.D, p.in_dtype, p.a, 1.0f, state, nseg, cps};
    rc = launch_scan_t<TIN, SCAN_DV>(dv, BH, a.stream);
    if (rc) return rc;
    // dK: x = v, y = grad_o (scaled by w), z = q
    ScanParams dk{a.v, a.grad_o, a.q, a.vs, a.gos, a.qs, a.dk, wbuf, ebuf, p.H, p.Nq, p.D, p.in_dtype, 1.0f, p.a, state, nseg, cps};
    return launch_scan_t<TIN, SCAN_DK>(dk, BH, a.stream);
}

int launch_bwd_scan(const BwdArgs& a) {
    if (!scan_bwd_supported(a.prob)) return FASTMAX_E_BAD_SHAPE;
    if (!a.workspace || a.workspace_bytes < scan_bwd_workspace(a.prob)) return FASTMAX_E_WORKSPACE;
    if (a.prob.in_dtype == FASTMAX_F32) return launch_scan_bwd_t<float>(a);
    return launch_scan_bwd_t<f16_t>(a);
}

}  // namespace fastmax
