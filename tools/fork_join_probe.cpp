// fork_join_probe.cpp -- what does a cross-queue dependency cost inside a captured hipGraph?  (round 6, forward_2d3d at B = 1)
// A weight prefetch for launch k + 1 that runs BESIDE launch k needs a second branch in the captured forward: fork in front
// of launch k, join in front of launch k + 1.  This probe replays a chain of 60 dependent launches of a ~3 us kernel
//   (a) as a plain chain,
//   (b) with a side-branch kernel beside every launch (fork: event record + wait on the side stream; join: the chain's next
//       launch waits for the side branch's event),
// from a hipGraph, and prints the time per chain step.  (b) - (a) is what the fork + join add to a dependent launch; the
// prefetch can save at most 0.8 us per launch on average (tools/kbench_enc.bin 1 --each --dup: 36.6 us over 45 launches).
// Build: hipcc --offload-arch=gfx950 -O3 tools/fork_join_probe.cpp -o tools/fork_join_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void spin_kernel(float* p, int iters)
{
    float x = p[threadIdx.x & 63];
    for (int i = 0; i < iters; ++i) x = x * 1.0000001f + 1e-9f;
    if (x == 12345.678f) p[0] = x;
}

__global__ void touch_kernel(const float* __restrict__ w, size_t n, float* sink)
{
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += w[i * 32];
    if (acc == 12345.678f) sink[0] = acc;
}

static double replay(hipStream_t s, hipStream_t side, float* buf, const float* w, int steps, bool fork, int spin)
{
    hipEvent_t ev_fork[64], ev_join[64];
    for (int i = 0; i < steps; ++i) { CK(hipEventCreateWithFlags(&ev_fork[i], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ev_join[i], hipEventDisableTiming)); }
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int k = 0; k < steps; ++k) {
        if (fork) {
            CK(hipEventRecord(ev_fork[k], s));
            CK(hipStreamWaitEvent(side, ev_fork[k], 0));
            hipLaunchKernelGGL(touch_kernel, dim3(128), dim3(256), 0, side, w, (size_t)1 << 17, buf + 64);  // 16 MB touched line by line
            CK(hipEventRecord(ev_join[k], side));
        }
        hipLaunchKernelGGL(spin_kernel, dim3(128), dim3(512), 0, s, buf, spin);
        if (fork) CK(hipStreamWaitEvent(s, ev_join[k], 0));
    }
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 20; ++i) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double best = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < 100; ++i) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms * 1e3 / 100 < best) best = ms * 1e3 / 100;
    }
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    return best / steps;
}

int main()
{
    hipStream_t s, side; CK(hipStreamCreate(&s)); CK(hipStreamCreate(&side));
    float *buf, *w; CK(hipMalloc(&buf, 4096)); CK(hipMemset(buf, 0, 4096)); CK(hipMalloc(&w, (size_t)16 << 20)); CK(hipMemset(w, 0, (size_t)16 << 20));
    const int steps = 60;
    for (int spin : {200, 1500, 4000}) {
        const double a = replay(s, side, buf, w, steps, false, spin), b = replay(s, side, buf, w, steps, true, spin);
        printf("chain of %d dependent launches, spin %4d: plain %.2f us per step; with a forked side launch beside every step %.2f us per step (+%.2f)\n",
               steps, spin, a, b, b - a);
    }
    return 0;
}
