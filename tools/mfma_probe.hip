// Standalone probe of the gfx950 primitives the fastmax MFMA kernel relies on.  Exact integer data,
// host-checked.  Build: hipcc --offload-arch=gfx950 -O2 tools/mfma_probe.hip -o tools/mfma_probe.bin
//  T1  v_mfma_f32_16x16x32_bf16 operand / result lane maps
//  T2  two 16x16 f32 accumulator tiles reused as the A operand of the next MFMA (X^T . B) with the
//      permuted k order
//  T3  ds_read_b64_tr_b16: which element lands in which lane
//  T4  O^T = V^T . P^T with V^T fragments from ds_read_b64_tr_b16 of a row-major V image and P^T taken
//      from accumulator-layout registers (B operand, permuted k)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __bf16 tobf(float x) { return (__bf16)x; }

// ---------------- T1
__global__ void t1_kernel(const float* A /*16x32*/, const float* B /*32x16*/, float* C /*16x16*/) {
    const int l = threadIdx.x, r = l & 15, q = l >> 4;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = tobf(A[r * 32 + 8 * q + j]);
        b[j] = tobf(B[(8 * q + j) * 16 + r]);
    }
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    for (int i = 0; i < 4; ++i) C[(4 * q + i) * 16 + r] = c[i];
}

// ---------------- T2: Y[j'][i] = sum_{m<32} X[m][j'] * Q[i][m]
__global__ void t2_kernel(const float* X /*32x16: [m][j']*/, const float* Q /*16x32: [i][m]*/, float* Y /*16x16*/) {
    const int l = threadIdx.x, r = l & 15, q = l >> 4;
    // accumulator layout of tile t: lane holds X[16t + 4q + reg][r]
    f32x4 x0, x1;
    for (int i = 0; i < 4; ++i) { x0[i] = X[(4 * q + i) * 16 + r]; x1[i] = X[(16 + 4 * q + i) * 16 + r]; }
    bf16x8 a, b;
    for (int j = 0; j < 4; ++j) { a[j] = tobf(x0[j]); a[4 + j] = tobf(x1[j]); }
    for (int j = 0; j < 8; ++j) {
        const int m = (j < 4) ? 4 * q + j : 16 + 4 * q + (j - 4);
        b[j] = tobf(Q[r * 32 + m]);
    }
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    for (int i = 0; i < 4; ++i) Y[(4 * q + i) * 16 + r] = c[i];
}

// ---------------- T3: raw dump of ds_read_b64_tr_b16.  LDS tile [16 rows][32 cols] bf16-sized shorts,
// value = row*100+col.  Group g (lanes 16g..16g+15) addresses the block rows 4g..4g+3, cols 16..31:
// lane 4q'+p' of the group supplies &tile[4g+q'][16+4p'].
__global__ void t3_kernel(short* out /*64 x 4*/) {
    __shared__ __attribute__((aligned(16))) short tile[16 * 32];
    const int l = threadIdx.x;
    for (int i = l; i < 16 * 32; i += 64) tile[i] = (short)((i / 32) * 100 + (i % 32));
    __syncthreads();
    const int g = l >> 4, qq = (l & 15) >> 2, pp = l & 3;
    const short* src = &tile[(4 * g + qq) * 32 + 16 + 4 * pp];
    s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)src);
    for (int i = 0; i < 4; ++i) out[l * 4 + i] = t[i];
}

// ---------------- T4: O[d][i] = sum_{j<32} V[j][d] * P[j][i],  d<16, i<16
__global__ void t4_kernel(const float* V /*32x16 [j][d]*/, const float* P /*32x16 [j][i]*/, float* O /*16x16 [d][i]*/) {
    __shared__ __attribute__((aligned(16))) __bf16 vimg[32 * 16];       // row-major [j][d], row stride 16
    const int l = threadIdx.x, r = l & 15, q = l >> 4;
    for (int i = l; i < 32 * 16; i += 64) vimg[i] = tobf(V[i]);
    __syncthreads();
    // A = V^T: lane (row d = r, k-group q) needs V[4q+0..3][d] and V[16+4q+0..3][d] (permuted k order that
    // matches the accumulator-layout B operand).  tr read: lane 4q'+p' of group q supplies row 4q+q', cols 4p'
    const int qq = (l & 15) >> 2, pp = l & 3;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)&vimg[(4 * q + qq) * 16 + 4 * pp]);
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)&vimg[(16 + 4 * q + qq) * 16 + 4 * pp]);
    union { bf16x8 v; s16x4 h[2]; } a;
    a.h[0] = lo; a.h[1] = hi;
    // B = P^T taken from the accumulator layout of S^T tiles: lane holds P[16t + 4q + reg][i = r]
    bf16x8 b;
    for (int j = 0; j < 4; ++j) { b[j] = tobf(P[(4 * q + j) * 16 + r]); b[4 + j] = tobf(P[(16 + 4 * q + j) * 16 + r]); }
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b, c, 0, 0, 0);
    for (int i = 0; i < 4; ++i) O[(4 * q + i) * 16 + r] = c[i];
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 2; } } while (0)

int main() {
    int fails = 0;
    float *dA, *dB, *dC;
    CK(hipMalloc(&dA, 4096)); CK(hipMalloc(&dB, 4096)); CK(hipMalloc(&dC, 4096));
    std::vector<float> A(512), B(512), C(256), R(256);
    auto fill = [](std::vector<float>& x, int seed) { for (size_t i = 0; i < x.size(); ++i) x[i] = (float)(((i * 7 + seed * 13) % 11) - 5); };
    // T1
    fill(A, 1); fill(B, 2);
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { float s = 0; for (int k = 0; k < 32; ++k) s += A[i * 32 + k] * B[k * 16 + j]; R[i * 16 + j] = s; }
    CK(hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 2048, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(t1_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dC); CK(hipDeviceSynchronize());
    CK(hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost));
    { int bad = 0; for (int i = 0; i < 256; ++i) bad += (C[i] != R[i]); printf("T1 mfma 16x16x32 lane maps: %s (%d mismatches)\n", bad ? "FAIL" : "PASS", bad); fails += bad != 0; }
    // T2
    fill(A, 3); fill(B, 4);   // A = X[32][16], B = Q[16][32]
    for (int jp = 0; jp < 16; ++jp) for (int i = 0; i < 16; ++i) { float s = 0; for (int m = 0; m < 32; ++m) s += A[m * 16 + jp] * B[i * 32 + m]; R[jp * 16 + i] = s; }
    CK(hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 2048, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(t2_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dC); CK(hipDeviceSynchronize());
    CK(hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost));
    { int bad = 0; for (int i = 0; i < 256; ++i) bad += (C[i] != R[i]); printf("T2 accumulator tiles as A operand (X^T.B): %s (%d mismatches)\n", bad ? "FAIL" : "PASS", bad); fails += bad != 0; }
    // T3
    short* dS; CK(hipMalloc(&dS, 64 * 4 * 2));
    hipLaunchKernelGGL(t3_kernel, dim3(1), dim3(64), 0, 0, dS); CK(hipDeviceSynchronize());
    std::vector<short> S(256);
    CK(hipMemcpy(S.data(), dS, 512, hipMemcpyDeviceToHost));
    { int bad = 0;
      for (int l = 0; l < 64; ++l) for (int e = 0; e < 4; ++e) { int g = l >> 4, i = l & 15; int exp = (4 * g + e) * 100 + 16 + i; bad += (S[l * 4 + e] != exp); }
      printf("T3 ds_read_b64_tr_b16 (lane i of group g gets column i of rows 4g..4g+3): %s (%d mismatches)\n", bad ? "FAIL" : "PASS", bad);
      if (bad) for (int l = 0; l < 64; ++l) printf("  lane %2d: %5d %5d %5d %5d\n", l, S[l * 4], S[l * 4 + 1], S[l * 4 + 2], S[l * 4 + 3]);
      fails += bad != 0; }
    // T4
    fill(A, 5); fill(B, 6);   // A = V[32][16], B = P[32][16]
    for (int d = 0; d < 16; ++d) for (int i = 0; i < 16; ++i) { float s = 0; for (int j = 0; j < 32; ++j) s += A[j * 16 + d] * B[j * 16 + i]; R[d * 16 + i] = s; }
    CK(hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 2048, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(t4_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dC); CK(hipDeviceSynchronize());
    CK(hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost));
    { int bad = 0; for (int i = 0; i < 256; ++i) bad += (C[i] != R[i]); printf("T4 V^T (tr read) x P^T (accumulator layout): %s (%d mismatches)\n", bad ? "FAIL" : "PASS", bad); fails += bad != 0; }
    printf("probe: %s\n", fails ? "FAILED" : "ALL PASS");
    return fails ? 1 : 0;
}
