// launch_floor.cpp -- what ONE dependent launch costs inside a replayed hipGraph on this GPU, by kernel shape.
// The B = 1 encoder is a chain of ~50 dependent launches whose arithmetic is a few microseconds in all; this
// measures the floor that chain sits on.  Build: hipcc --offload-arch=gfx950 -O3 tools/launch_floor.cpp -o /tmp/launch_floor
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_empty(const float*, float*, int) {}

// one dependent round trip: every thread reads what the previous launch wrote and writes for the next
__global__ void k_touch(const float* in, float* out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] + 1.0f;
}

// two dependent round trips (load -> address -> load), the shape of "statistics, then normalise"
__global__ void k_touch2(const float* in, float* out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const float a = in[i];
        const int j = ((int)a & 1023) ^ (i & (n - 1));
        out[i] = in[j] + a;
    }
}

// as k_touch, but the value comes from the NEXT workgroup's slice: written through another XCD's L2 by the previous launch
__global__ void k_touch_far(const float* in, float* out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[(i + blockDim.x) & (n - 1)] + 1.0f;
}

// HOPS dependent loads, each from another workgroup's slice
template <int HOPS>
__global__ void k_chase_far(const float* in, float* out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    int j = (i + blockDim.x) & (n - 1);
    float a = 0.0f;
#pragma unroll
    for (int h = 0; h < HOPS; ++h) {
        a += in[j];
        j = (j + blockDim.x * (1 + ((int)a & 1))) & (n - 1);
    }
    out[i] = a;
}

// a long straight-line body (about 4 * N instructions) in front of the store: instruction fetch of a cold kernel
template <int N>
__global__ void k_fat(const float* in, float* out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float a = in[i & (n - 1)], b = a + 1.0f, c = a + 2.0f, d = a + 3.0f;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        a = __builtin_fmaf(a, 1.0001f, (float)k);
        b = __builtin_fmaf(b, 0.9999f, a);
        c = __builtin_fmaf(c, 1.0002f, b);
        d = __builtin_fmaf(d, 0.9998f, c);
    }
    out[i & (n - 1)] = a + b + c + d;
}

// big static LDS + many registers per thread
__global__ __launch_bounds__(512) void k_big(const float* in, float* out, int n)
{
    __shared__ float lds[32768];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    lds[threadIdx.x] = in[i & (n - 1)];
    __syncthreads();
    out[i & (n - 1)] = lds[threadIdx.x ^ 1];
}

// streams `bytes_per_wg` of weights per workgroup (float4 per thread per step), all loads in flight at once
template <int STEPS>
__global__ void k_stream(const float* in, float* out, int n, const float4* w)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float4 acc[STEPS];
#pragma unroll
    for (int s = 0; s < STEPS; ++s) acc[s] = w[(size_t)s * gridDim.x * blockDim.x + i];
    float v = in[i & (n - 1)];
#pragma unroll
    for (int s = 0; s < STEPS; ++s) v += acc[s].x + acc[s].y + acc[s].z + acc[s].w;
    out[i & (n - 1)] = v;
}

// The same bytes as k_stream, fetched the way an MFMA operand fragment is when the matrix is row-major in memory:
// lane (r = lane % 16, q = lane / 16) reads the float4 at row r, column 4 q of a [16][ld] tile, so the four lanes of a
// quad touch four different 128-byte lines.
template <int STEPS>
__global__ void k_stream_frag(const float* in, float* out, int n, const float4* w, int ld4)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // the workgroup's 8 waves x STEPS steps cover a [16][8 * STEPS * 4 float4] tile per 16 rows; tile base per workgroup
    const float4* base = w + (size_t)blockIdx.x * 16 * ld4 + (size_t)(lane & 15) * ld4 + (lane >> 4);
    float4 acc[STEPS];
#pragma unroll
    for (int s = 0; s < STEPS; ++s) acc[s] = base[(wave * STEPS + s) * 4];
    float v = in[i & (n - 1)];
#pragma unroll
    for (int s = 0; s < STEPS; ++s) v += acc[s].x + acc[s].y + acc[s].z + acc[s].w;
    out[i & (n - 1)] = v;
}

// Every workgroup re-reads the SAME 128 KB ([64][512] floats, L2 resident), as the column-tile workgroups of a skinny
// GEMM re-read its activations: FRAG = 0 row-contiguous (a wave reads 1 KB of one row), FRAG = 1 as MFMA fragments
// (lane = (row % 16, 4-float group): the four lanes of a quad sit in four different 128-byte lines).
template <int FRAG>
__global__ __launch_bounds__(512) void k_reread(const float* in, float* out, int n, const float4* x)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float4 acc[16];
    if (FRAG) {
        // wave = K slice of 64 floats; step st = 16 floats; row tile rt = 16 rows
#pragma unroll
        for (int st = 0; st < 4; ++st)
#pragma unroll
            for (int rt = 0; rt < 4; ++rt)
                acc[st * 4 + rt] = x[(size_t)(rt * 16 + (lane & 15)) * 128 + wave * 16 + st * 4 + (lane >> 4)];
    } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = x[(size_t)(wave * 8 + (j >> 1)) * 128 + (j & 1) * 64 + lane];
    }
    float v = in[threadIdx.x];
#pragma unroll
    for (int j = 0; j < 16; ++j) v += acc[j].x + acc[j].y + acc[j].z + acc[j].w;
    out[(blockIdx.x * blockDim.x + threadIdx.x) & (n - 1)] = v;
}

template <typename F>
static int chain(const char* name, int launches, F launch, hipStream_t s)
{
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < launches; ++i) launch(i);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    const int reps = 50;
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("{\"chain\": \"%s\", \"launches\": %d, \"us_per_launch\": %.3f}\n", name, launches, ms * 1e3 / reps / launches);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    return 0;
}

int main()
{
    hipStream_t s; CK(hipStreamCreate(&s));
    const int n = 1 << 16;
    float *a, *b; float4* w;
    CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4));
    const size_t wbytes = (size_t)64 << 20;
    CK(hipMalloc(&w, wbytes));
    CK(hipMemset(a, 0, n * 4)); CK(hipMemset(b, 0, n * 4)); CK(hipMemset(w, 0, wbytes));
    const int L = 64;
    auto pp = [&](int i, const float** in, float** out) { *in = (i & 1) ? b : a; *out = (i & 1) ? a : b; };
    for (int wg : {16, 256}) {
        for (int th : {256, 512}) {
            char nm[96];
            snprintf(nm, sizeof nm, "empty %d x %d", wg, th);
            if (chain(nm, L, [&](int i) { const float* in; float* out; pp(i, &in, &out); hipLaunchKernelGGL(k_empty, dim3(wg), dim3(th), 0, s, in, out, n); }, s)) return 1;
            snprintf(nm, sizeof nm, "load-store %d x %d", wg, th);
            if (chain(nm, L, [&](int i) { const float* in; float* out; pp(i, &in, &out); hipLaunchKernelGGL(k_touch, dim3(wg), dim3(th), 0, s, in, out, n); }, s)) return 1;
            snprintf(nm, sizeof nm, "load-load-store %d x %d", wg, th);
            if (chain(nm, L, [&](int i) { const float* in; float* out; pp(i, &in, &out); hipLaunchKernelGGL(k_touch2, dim3(wg), dim3(th), 0, s, in, out, n); }, s)) return 1;
        }
    }
    for (int wg : {16, 128}) {
        char nm[96];
        snprintf(nm, sizeof nm, "far load-store %d x 512", wg);
        if (chain(nm, L, [&](int i) { const float* in; float* out; pp(i, &in, &out); hipLaunchKernelGGL(k_touch_far, dim3(wg), dim3(512), 0, s, in, out, n); }, s)) return 1;
        snprintf(nm, sizeof nm, "far chase x2 %d x 512", wg);
        if (chain(nm, L, [&](int i) { const float* in; float* out; pp(i, &in, &out); hipLaunchKernelGGL(k_chase_far<2>, dim3(wg), dim3(512), 0, s, in, out, n); }, s)) return 1;
        snprintf(nm, sizeof nm, "far chase x4 %d x 512", wg);
        if (chain(nm, L, [&](int i) { const float* in; float* out; pp(i, &in, &out); hipLaunchKernelGGL(k_chase_far<4>, dim3(wg), dim3(512), 0, s, in, out, n); }, s)) return 1;
        snprintf(nm, sizeof nm, "far chase x8 %d x 512", wg);
        if (chain(nm, L, [&](int i) { const float* in; float* out; pp(i, &in, &out); hipLaunchKernelGGL(k_chase_far<8>, dim3(wg), dim3(512), 0, s, in, out, n); }, s)) return 1;
    }
    if (chain("fat code 1k fma x4, 128 x 512", L, [&](int i) { const float* in; float* out; pp(i, &in, &out); hipLaunchKernelGGL(k_fat<1024>, dim3(128), dim3(512), 0, s, in, out, n); }, s)) return 1;
    if (chain("fat code 4k fma x4, 128 x 512", L, [&](int i) { const float* in; float* out; pp(i, &in, &out); hipLaunchKernelGGL(k_fat<4096>, dim3(128), dim3(512), 0, s, in, out, n); }, s)) return 1;
    if (chain("128 KB LDS, 128 x 512", L, [&](int i) { const float* in; float* out; pp(i, &in, &out); hipLaunchKernelGGL(k_big, dim3(128), dim3(512), 0, s, in, out, n); }, s)) return 1;
    // weight streaming: 256 workgroups x 512 threads x STEPS float4 = 2 MB x STEPS per launch, a different 2*STEPS MB each launch
    if (chain("stream 2 MB (256 x 512)", 16, [&](int i) { const float* in; float* out; pp(i, &in, &out); hipLaunchKernelGGL(k_stream<1>, dim3(256), dim3(512), 0, s, in, out, n, w + (size_t)i * 131072 * 1); }, s)) return 1;
    if (chain("stream 8 MB (256 x 512)", 8, [&](int i) { const float* in; float* out; pp(i, &in, &out); hipLaunchKernelGGL(k_stream<4>, dim3(256), dim3(512), 0, s, in, out, n, w + (size_t)i * 131072 * 4); }, s)) return 1;
    for (int wg : {32, 128, 256}) {
        char nm[96];
        snprintf(nm, sizeof nm, "re-read 128 KB per workgroup, row-contiguous, %d x 512", wg);
        if (chain(nm, 16, [&](int i) { const float* in; float* out; pp(i, &in, &out); hipLaunchKernelGGL(k_reread<0>, dim3(wg), dim3(512), 0, s, in, out, n, w); }, s)) return 1;
        snprintf(nm, sizeof nm, "re-read 128 KB per workgroup, as MFMA fragments, %d x 512", wg);
        if (chain(nm, 16, [&](int i) { const float* in; float* out; pp(i, &in, &out); hipLaunchKernelGGL(k_reread<1>, dim3(wg), dim3(512), 0, s, in, out, n, w); }, s)) return 1;
    }
    {   // two independent dependent chains forked onto two streams inside ONE captured graph: do the branches overlap?
        hipStream_t s1; CK(hipStreamCreate(&s1));
        hipEvent_t fork, join; CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
        float *c, *d; CK(hipMalloc(&c, n * 4)); CK(hipMalloc(&d, n * 4)); CK(hipMemset(c, 0, n * 4)); CK(hipMemset(d, 0, n * 4));
        for (int kind = 0; kind < 3; ++kind) {
            for (int lanes = 1; lanes <= 2; ++lanes) {
                char nm[128];
                snprintf(nm, sizeof nm, "%s, %d chain(s) of 64 in one graph (us per launch of ONE chain)",
                         kind == 0 ? "load-store 128 x 512" : kind == 1 ? "stream 2 MB, 128 x 512" : "far chase x8 128 x 512", lanes);
                auto body = [&](hipStream_t st, float* x, float* y, int i, int lane) {
                    const float* in = (i & 1) ? y : x; float* out = (i & 1) ? x : y;
                    if (kind == 0) hipLaunchKernelGGL(k_touch, dim3(128), dim3(512), 0, st, in, out, n);
                    else if (kind == 1) hipLaunchKernelGGL(k_stream<2>, dim3(128), dim3(512), 0, st, in, out, n, w + (size_t)((i * 2 + lane) & 31) * 131072);
                    else hipLaunchKernelGGL(k_chase_far<8>, dim3(128), dim3(512), 0, st, in, out, n);
                };
                hipGraph_t g; hipGraphExec_t ge;
                CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
                if (lanes == 2) { CK(hipEventRecord(fork, s)); CK(hipStreamWaitEvent(s1, fork, 0)); }
                for (int i = 0; i < 64; ++i) {
                    body(s, a, b, i, 0);
                    if (lanes == 2) body(s1, c, d, i, 1);
                }
                if (lanes == 2) { CK(hipEventRecord(join, s1)); CK(hipStreamWaitEvent(s, join, 0)); }
                CK(hipStreamEndCapture(s, &g));
                CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
                hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
                for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(ge, s));
                CK(hipStreamSynchronize(s));
                CK(hipEventRecord(e0, s));
                for (int i = 0; i < 50; ++i) CK(hipGraphLaunch(ge, s));
                CK(hipEventRecord(e1, s));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                printf("{\"chain\": \"%s\", \"launches\": %d, \"us_per_launch\": %.3f}\n", nm, 64 * lanes, ms * 1e3 / 50 / 64);
                CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
            }
        }
    }
    {   // host side: what issuing one launch costs the CPU (eager mode), by launch API
        auto now = [] { return std::chrono::steady_clock::now(); };
        const int reps = 2000;
        CK(hipStreamSynchronize(s));
        auto t0 = now();
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_touch, dim3(16), dim3(256), 0, s, a, b, n);
        auto t1 = now();
        CK(hipStreamSynchronize(s));
        printf("{\"host\": \"hipLaunchKernelGGL, small kernarg\", \"us_per_launch_issue\": %.3f}\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / reps);
        hipFunction_t fn;
        CK(hipGetFuncBySymbol(&fn, reinterpret_cast<const void*>(k_touch)));
        const float* pa = a; float* pb = b; int nn = n;
        void* args[3] = {&pa, &pb, &nn};
        t0 = now();
        for (int i = 0; i < reps; ++i) (void)hipModuleLaunchKernel(fn, 16, 1, 1, 256, 1, 1, 0, s, args, nullptr);
        t1 = now();
        CK(hipStreamSynchronize(s));
        printf("{\"host\": \"hipModuleLaunchKernel, cached function\", \"us_per_launch_issue\": %.3f}\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / reps);
    }
    // fragment-order fetch of the same 8 MB: each workgroup owns 16 rows of a [4096][ld] matrix, ld = 8 waves x 4 steps x 16 floats
    if (chain("stream 8 MB as MFMA fragments of a row-major matrix", 8, [&](int i) { const float* in; float* out; pp(i, &in, &out); hipLaunchKernelGGL(k_stream_frag<4>, dim3(256), dim3(512), 0, s, in, out, n, w + (size_t)i * 131072 * 4, 8 * 4 * 4); }, s)) return 1;
    if (chain("stream 16 MB as MFMA fragments of a row-major matrix", 4, [&](int i) { const float* in; float* out; pp(i, &in, &out); hipLaunchKernelGGL(k_stream_frag<8>, dim3(256), dim3(512), 0, s, in, out, n, w + (size_t)i * 131072 * 8, 8 * 8 * 4); }, s)) return 1;
    if (chain("stream 16 MB (256 x 512)", 4, [&](int i) { const float* in; float* out; pp(i, &in, &out); hipLaunchKernelGGL(k_stream<8>, dim3(256), dim3(512), 0, s, in, out, n, w + (size_t)i * 131072 * 8); }, s)) return 1;
    return 0;
}
