// What can a trivial kernel with the headline operator's traffic mix reach on this GPU?  Reads three fp32 streams of 512 MiB,
// writes one (o = q + k + v), 16 bytes per lane, non-temporal, and also a plain copy (1 read : 1 write) for reference.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/hbm_mix_probe.bin tools/hbm_mix_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int NT>
__global__ __launch_bounds__(256) void mix3(const f4* __restrict__ a, const f4* __restrict__ b, const f4* __restrict__ c, f4* __restrict__ o, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        f4 x, y, z;
        if (NT) { x = __builtin_nontemporal_load(a + i); y = __builtin_nontemporal_load(b + i); z = __builtin_nontemporal_load(c + i); }
        else { x = a[i]; y = b[i]; z = c[i]; }
        f4 r = x + y + z;
        if (NT) __builtin_nontemporal_store(r, o + i); else o[i] = r;
    }
}
template <int NT>
__global__ __launch_bounds__(256) void copy1(const f4* __restrict__ a, f4* __restrict__ o, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        f4 x = NT ? __builtin_nontemporal_load(a + i) : a[i];
        if (NT) __builtin_nontemporal_store(x, o + i); else o[i] = x;
    }
}
// the operator's shape of access: one workgroup per 1 MiB "head", walking it in 16 KiB chunks, 3 streams in, 1 out
template <int NT>
__global__ __launch_bounds__(256) void heads3(const f4* __restrict__ a, const f4* __restrict__ b, const f4* __restrict__ c, f4* __restrict__ o, int chunks) {
    const size_t base = (size_t)blockIdx.x * chunks * 1024;                 // f4 units: 16 KiB = 1024 f4
    for (int ch = 0; ch < chunks; ++ch)
        for (int p = 0; p < 4; ++p) {
            const size_t i = base + (size_t)ch * 1024 + p * 256 + threadIdx.x;
            f4 x, y, z;
            if (NT) { x = __builtin_nontemporal_load(a + i); y = __builtin_nontemporal_load(b + i); z = __builtin_nontemporal_load(c + i); }
            else { x = a[i]; y = b[i]; z = c[i]; }
            f4 r = x + y + z;
            if (NT) __builtin_nontemporal_store(r, o + i); else o[i] = r;
        }
}
int main() {
    const size_t bytes = 512ull << 20, n = bytes / 16;
    f4 *a, *b, *c, *o;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&c, bytes); hipMalloc(&o, bytes);
    hipMemset(a, 0, bytes); hipMemset(b, 0, bytes); hipMemset(c, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, double traffic, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
        printf("%-44s %7.1f us  %6.2f TB/s\n", name, ms * 1e3, traffic / (ms * 1e-3) / 1e12);
    };
    for (int grid : {2048, 8192, 32768}) {
        char nm[96];
        snprintf(nm, sizeof nm, "3 reads + 1 write, grid-stride, nt, grid %d", grid);
        run(nm, 4.0 * bytes, [&] { hipLaunchKernelGGL(mix3<1>, dim3(grid), dim3(256), 0, 0, a, b, c, o, n); });
    }
    run("3 reads + 1 write, grid-stride, default policy", 4.0 * bytes, [&] { hipLaunchKernelGGL(mix3<0>, dim3(8192), dim3(256), 0, 0, a, b, c, o, n); });
    run("3 reads + 1 write, 512 WGs x 64 chunks, nt", 4.0 * bytes, [&] { hipLaunchKernelGGL(heads3<1>, dim3(512), dim3(256), 0, 0, a, b, c, o, 64); });
    run("3 reads + 1 write, 512 WGs x 64 chunks, default", 4.0 * bytes, [&] { hipLaunchKernelGGL(heads3<0>, dim3(512), dim3(256), 0, 0, a, b, c, o, 64); });
    for (int wgs : {256, 1024, 2048, 4096}) {
        char nm[96];
        snprintf(nm, sizeof nm, "3 reads + 1 write, %d WGs x %d chunks, nt", wgs, 32768 / wgs);
        run(nm, 4.0 * bytes, [&] { hipLaunchKernelGGL(heads3<1>, dim3(wgs), dim3(256), 0, 0, a, b, c, o, 32768 / wgs); });
    }
    run("copy 1 read + 1 write, nt", 2.0 * bytes, [&] { hipLaunchKernelGGL(copy1<1>, dim3(8192), dim3(256), 0, 0, a, o, n); });
    run("copy 1 read + 1 write, default policy", 2.0 * bytes, [&] { hipLaunchKernelGGL(copy1<0>, dim3(8192), dim3(256), 0, 0, a, o, n); });
    return 0;
}
