// Probe: cost of LDS atomics per wave-instruction (conflict-free addresses), 4 waves per CU, every CU busy.
// Build: hipcc --offload-arch=gfx950 -O3 tools/lds_atomic_probe.cpp -o tools/lds_atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters)
{
    __shared__ __attribute__((aligned(16))) float buf[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) buf[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* p = buf + wave * 2048 + lane;  // per-wave region, lane-linear: conflict-free
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            float* q = p + 64 * j;
            if (MODE == 0) __hip_atomic_fetch_add(q, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (MODE == 1) __hip_atomic_fetch_add(reinterpret_cast<unsigned*>(q), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (MODE == 2) { *q = *q + 1.0f; }
            if (MODE == 3) __hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(__builtin_assume_aligned(buf + wave * 2048 + 2 * lane + 128 * (j & 7), 8)), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (MODE == 2) asm volatile("" ::: "memory");
    }
    __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = buf[threadIdx.x];
}
int main()
{
    float* d; hipMalloc(&d, 256 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    const char* names[4] = {"ds_add_f32", "ds_add_u32", "read+add+write", "ds_add_u64"};
    for (int mode = 0; mode < 4; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, d, iters);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, d, iters);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, d, iters);
            if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(256), 0, 0, d, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("%-16s %.3f ms -> %.1f cycles (2.4 GHz) per wave-instruction per wave\n", names[mode], ms, ms * 1e-3 * 2.4e9 / (iters * 16.0));
        }
    }
    return 0;
}
