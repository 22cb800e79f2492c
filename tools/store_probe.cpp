// store-pattern probe: N regions of 32 KiB, one wave per region, persistent grid; which pattern reaches what write rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int P, bool NT>
__global__ __launch_bounds__(256) void probe(float* out, long N, int lds_touch)
{
    extern __shared__ float lds[];
    if (lds_touch < 0) lds[threadIdx.x] = 1.0f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long nstep = (long)gridDim.x * 4;
    const int e = lane & 7, a0 = (lane >> 3) & 1, bq = lane >> 4;
    for (long n = (long)blockIdx.x * 4 + wave; n < N; n += nstep) {
        float* o = out + n * 8192;
        const float v = (float)n;
        auto st = [&](float* p, float x) { if (NT) __builtin_nontemporal_store(x, p); else *p = x; };
        auto st4 = [&](float* p, float x) { f32x4 t = {x, x, x, x}; if (NT) __builtin_nontemporal_store(t, (f32x4*)p); else *(f32x4*)p = t; };
        if (P == 0) {  // shipped: quarter, pass, 16 channel stores of two 128-byte runs
            for (int q = 0; q < 4; ++q)
                for (int p = 0; p < 2; ++p)
#pragma unroll
                    for (int c = 0; c < 16; ++c) st(o + c * 512 + q * 128 + a0 * 64 + (4 * p + bq) * 8 + e, v);
        } else if (P == 1) {  // fill-like: 16 bytes per lane, contiguous
#pragma unroll 8
            for (int i = 0; i < 32; ++i) st4(o + i * 256 + lane * 4, v);
        } else if (P == 2) {  // 4 bytes per lane, contiguous
#pragma unroll 16
            for (int i = 0; i < 128; ++i) st(o + i * 64 + lane, v);
        } else if (P == 3) {  // round-5 first version: per quarter 8 x (2 channels x 512 contiguous bytes), 16 bytes per lane
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < 8; ++i) st4(o + (2 * i + (lane >> 5)) * 512 + q * 128 + (lane & 31) * 4, v);
        } else if (P == 4) {  // per pass 4 x (4 channels x 2 runs of 128 bytes), 16 bytes per lane
            for (int q = 0; q < 4; ++q)
                for (int p = 0; p < 2; ++p)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        st4(o + (4 * i + (lane >> 4)) * 512 + q * 128 + ((lane >> 3) & 1) * 64 + p * 32 + (lane & 7) * 4, v);
        } else if (P == 5) {  // 4 bytes per lane, 256 contiguous bytes per channel and pass
            for (int q = 0; q < 4; ++q)
                for (int p = 0; p < 2; ++p)
#pragma unroll
                    for (int c = 0; c < 16; ++c) st(o + c * 512 + q * 128 + p * 64 + lane, v);
        } else if (P == 9) {  // P0 with the quarters in the order 0, 3, 1, 2 and the mirrored lane map in 3 and 2
            for (int qi = 0; qi < 4; ++qi) {
                const int q = (0x2130 >> (4 * qi)) & 3;
                const bool mir = qi & 1;
                for (int p = 0; p < 2; ++p)
#pragma unroll
                    for (int c = 0; c < 16; ++c)
                        st(o + c * 512 + q * 128 + (mir ? 127 - (a0 * 64 + (4 * p + bq) * 8 + e) : a0 * 64 + (4 * p + bq) * 8 + e), v);
            }
        } else if (P == 6) {  // channel-major: a whole 2 KiB plane at a time, 16 bytes per lane (needs the whole volume in LDS)
#pragma unroll
            for (int c = 0; c < 16; ++c) { st4(o + c * 512 + lane * 4, v); st4(o + c * 512 + 256 + lane * 4, v); }
        }
    }
}

// torch-like fill: a workgroup per contiguous 16 KiB chunk (PERSIST: chunks strided over a resident grid)
template <bool PERSIST, bool NT>
__global__ __launch_bounds__(256) void fill(float* out, long chunks)
{
    for (long ch = blockIdx.x; ch < chunks; ch += gridDim.x) {
        float* o = out + ch * 4096;
        const float v = (float)ch;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x4 t = {v, v, v, v};
            if (NT) __builtin_nontemporal_store(t, (f32x4*)(o + (i * 256 + threadIdx.x) * 4)); else *(f32x4*)(o + (i * 256 + threadIdx.x) * 4) = t;
        }
        if (!PERSIST) break;
    }
}
template <bool PERSIST, bool NT>
void run_fill(const char* name, float* out, long N)
{
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const long chunks = N * 2;
    const long grid = PERSIST ? prop.multiProcessorCount * 8 : chunks;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ts;
    for (int it = 0; it < 13; ++it) {
        CK(hipEventRecord(e0));
        fill<PERSIST, NT><<<(unsigned)grid, 256>>>(out, chunks);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (it >= 3) ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    printf("%-64s %s          min %.4f median %.4f ms -> %.2f TB/s\n", name, NT ? "nt " : "tmp", ts[0], ts[5], N * 32768.0 / ts[5] / 1e9);
}

template <int P, bool NT>
void run(const char* name, float* out, long N, int wg_per_cu, int lds_bytes)
{
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int grid = prop.multiProcessorCount * wg_per_cu;
    CK(hipFuncSetAttribute((const void*)probe<P, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ts;
    for (int it = 0; it < 13; ++it) {
        CK(hipEventRecord(e0));
        probe<P, NT><<<grid, 256, lds_bytes>>>(out, N, 0);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (it >= 3) ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    printf("%-64s %s wg/cu %d  min %.4f median %.4f ms -> %.2f TB/s\n", name, NT ? "nt " : "tmp", wg_per_cu, ts[0], ts[5], N * 32768.0 / ts[5] / 1e9);
}

int main()
{
    const long N = 200000;
    float* out; CK(hipMalloc(&out, N * 32768));
    CK(hipMemset(out, 0, N * 32768));
    for (int rep = 0; rep < 2; ++rep) {
        run<0, true>("P0 shipped: 4 B lanes, 2 x 128 B per instr, 2 KiB channel stride", out, N, 3, 48 * 1024);
        run<0, false>("P0", out, N, 3, 48 * 1024);
        run<1, true>("P1 fill-like: 16 B lanes, contiguous", out, N, 3, 48 * 1024);
        run<1, false>("P1", out, N, 3, 48 * 1024);
        run<2, true>("P2 4 B lanes, contiguous", out, N, 3, 48 * 1024);
        run<3, true>("P3 per quarter 8 x (2 ch x 512 B), 16 B lanes", out, N, 3, 48 * 1024);
        run<3, true>("P3 at 2 wg/cu", out, N, 2, 79 * 1024);
        run<4, true>("P4 per pass 4 x (4 ch x 2 x 128 B), 16 B lanes", out, N, 3, 48 * 1024);
        run<5, true>("P5 4 B lanes, 256 B per channel and pass", out, N, 3, 48 * 1024);
        run<6, true>("P6 plane at a time, 16 B lanes", out, N, 3, 48 * 1024);
        run<9, true>("P9 = P0 in quarter order 0,3,1,2, mirrored lanes", out, N, 3, 48 * 1024);
        run<9, false>("P9", out, N, 3, 48 * 1024);
        run_fill<false, false>("fill: one workgroup per 16 KiB chunk", out, N);
        run_fill<false, true>("fill: one workgroup per 16 KiB chunk", out, N);
        run_fill<true, false>("fill: persistent, chunks strided", out, N);
        run_fill<true, true>("fill: persistent, chunks strided", out, N);
        run<1, true>("P1 at 8 wg/cu", out, N, 8, 1024);
        run<0, true>("P0 at 8 wg/cu", out, N, 8, 1024);
        run<0, true>("P0 at 1 wg/cu", out, N, 1, 1024);
        run<1, true>("P1 at 1 wg/cu", out, N, 1, 1024);
    }
    return 0;
}
