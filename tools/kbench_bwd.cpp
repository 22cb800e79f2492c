// kbench_bwd.cpp -- developer micro-benchmark of the scorer's three backward kernels at the reference's training
// size (B = 12 per-sample rotation sets of N = 3000), kernel by kernel, with diagnostic switches
// (-DAHV_DIAG_NO_ATOMICS, -DAHV_DIAG_NO_DX: wrong results, to price a component).  Not part of the product.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I3dahv_amd/csrc -Iinclude tools/kbench_bwd.cpp -o tools/kbench_bwd
// (the kernels as the library launches them).  With -DAHV_BWD_DU_AMAX -o tools/kbench_bwd_atomics the head kernels also produce
// max |du| per sample and the LDS-atomic dV kernel of rounds 2-5 is timed and compared beside the read-modify-write one.
#include "../3dahv_amd/csrc/ahv_backward.hip"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

// the library's launcher (never called here) refers to the zero-fill launcher of ahv_ops.hip
namespace ahv { hipError_t launch_zero_fill(void* const*, const size_t*, int, hipStream_t) { return hipErrorNotSupported; } }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

int main(int argc, char** argv)
{
    const int B = argc > 1 ? atoi(argv[1]) : 12;
    const long N = argc > 2 ? atol(argv[2]) : 3000;
    const int iters = argc > 3 ? atoi(argv[3]) : 10;
    std::mt19937 rng(0);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> vol((size_t)B * 8192), ft((size_t)B * 2048), R((size_t)B * N * 9), W1(32 * 384), W2(1024), b2(32), gs((size_t)B * N);
    for (auto& x : vol) x = 1.15f * nd(rng);
    for (auto& x : ft) x = nd(rng) / 5.6f;
    for (auto& x : W1) x = nd(rng) * 0.03f;
    for (auto& x : W2) x = nd(rng) * 0.1f;
    for (auto& x : b2) x = nd(rng) * 0.1f;
    for (auto& x : gs) x = nd(rng);
    for (long n = 0; n < B * N; ++n) {
        double q[4], s = 0;
        for (double& c : q) { c = nd(rng); s += c * c; }
        const double t = 2.0 / s, r = q[0], i = q[1], j = q[2], k = q[3];
        const double m[9] = {1 - t * (j * j + k * k), t * (i * j - k * r), t * (i * k + j * r), t * (i * j + k * r), 1 - t * (i * i + k * k),
                             t * (j * k - i * r), t * (i * k - j * r), t * (j * k + i * r), 1 - t * (i * i + j * j)};
        for (int e = 0; e < 9; ++e) R[n * 9 + e] = (float)m[e];
    }
    float *dvol, *dft, *dR, *dW1, *dW2, *db2, *dgs, *dws, *gvol, *gvol2, *gft, *gW1, *gW2, *gb2, *dpart;
    unsigned* dmax;
    CK(hipMalloc(&dvol, vol.size() * 4)); CK(hipMalloc(&dft, ft.size() * 4)); CK(hipMalloc(&dR, R.size() * 4));
    CK(hipMalloc(&dW1, W1.size() * 4)); CK(hipMalloc(&dW2, W2.size() * 4)); CK(hipMalloc(&db2, b2.size() * 4));
    CK(hipMalloc(&dgs, gs.size() * 4)); CK(hipMalloc(&dws, (size_t)B * N * 2048 * 4)); CK(hipMalloc(&dmax, B * 4));
    CK(hipMalloc(&dpart, (size_t)1024 * 32 * 384 * 4)); CK(hipMalloc(&gvol, vol.size() * 4)); CK(hipMalloc(&gvol2, vol.size() * 4)); CK(hipMalloc(&gft, ft.size() * 4)); CK(hipMalloc(&gW1, W1.size() * 4));
    CK(hipMalloc(&gW2, W2.size() * 4)); CK(hipMalloc(&gb2, b2.size() * 4));
    CK(hipMemcpy(dvol, vol.data(), vol.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dft, ft.data(), ft.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dR, R.data(), R.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dW1, W1.data(), W1.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dW2, W2.data(), W2.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(db2, b2.data(), b2.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dgs, gs.data(), gs.size() * 4, hipMemcpyHostToDevice));
    int cu = 0;
    CK(hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, 0));
    int gy = B < cu ? B : cu, gx = cu / gy;
    const dim3 grid(gx, gy);
    hipEvent_t e[7];
    for (auto& x : e) CK(hipEventCreate(&x));
    double t[5] = {0, 0, 0, 0, 0};
    for (int it = -2; it < iters; ++it) {
        CK(hipMemsetAsync(gvol, 0, vol.size() * 4, 0)); CK(hipMemsetAsync(gft, 0, ft.size() * 4, 0));
        CK(hipMemsetAsync(gW1, 0, W1.size() * 4, 0)); CK(hipMemsetAsync(gW2, 0, 4096, 0)); CK(hipMemsetAsync(gb2, 0, 128, 0));
        CK(hipMemsetAsync(dmax, 0, B * 4, 0));
        CK(hipEventRecord(e[0], 0));
        hipLaunchKernelGGL(ahv::score_backward_head_kernel, grid, dim3(ahv::kBwdThreads), 0, 0, dvol, dft, dR, (long)(N * 9), dW1, dW2, db2,
                           B, N, dgs, dws, dmax, gft, gW2, gb2);
        CK(hipEventRecord(e[1], 0));
        hipLaunchKernelGGL(ahv::score_backward_w1_kernel, grid, dim3(ahv::kW1Threads), 0, 0, dvol, dR, (long)(N * 9), B, N, dws, dpart);
        hipLaunchKernelGGL(ahv::score_backward_w1_reduce_kernel, dim3(32 * 384 / 256, 16), dim3(256), 0, 0, dpart, gx * gy, gW1);
        CK(hipEventRecord(e[2], 0));
#ifdef AHV_BWD_DU_AMAX   // the LDS-atomic dV kernel of rounds 2-5 (needs max |du| per sample from the head kernel: kbench_bwd_atomics)
        hipLaunchKernelGGL(ahv::score_backward_volume_kernel, grid, dim3(ahv::kVolThreads), 0, 0, dR, (long)(N * 9), dW1, B, N, dws, dmax, gvol);
#endif
        CK(hipEventRecord(e[3], 0));
        // round 6: the same gradient without LDS atomics (private fp32 images, read-modify-write), into its own buffer
        CK(hipMemsetAsync(gvol2, 0, vol.size() * 4, 0));
        CK(hipEventRecord(e[4], 0));
        hipLaunchKernelGGL(ahv::score_backward_volume_rmw_kernel, grid, dim3(ahv::kRmwThreads), 0, 0, dR, (long)(N * 9), dW1, B, N, dws, gvol2);
        CK(hipEventRecord(e[5], 0));
        {   // the training pair's head kernel (u from the workspace; for the clock any 8 KB per hypothesis will do: du stands in)
            int gxs = cu / gy;
            if (gxs > (N + 7) / 8) gxs = (int)((N + 7) / 8);
            hipLaunchKernelGGL(ahv::score_backward_head_saved_kernel, dim3(gxs < 1 ? 1 : gxs, gy), dim3(ahv::kSavedThreads), 0, 0, dft, dW2, db2,
                               B, N, dgs, dws, dmax, gft, gW2, gb2);
        }
        CK(hipEventRecord(e[6], 0));
        CK(hipEventSynchronize(e[6]));
        CK(hipGetLastError());
        if (it >= 0) {
            for (int k = 0; k < 3; ++k) { float ms; CK(hipEventElapsedTime(&ms, e[k], e[k + 1])); t[k] += ms; }
            float ms; CK(hipEventElapsedTime(&ms, e[4], e[5])); t[3] += ms;
            CK(hipEventElapsedTime(&ms, e[5], e[6])); t[4] += ms;
        }
    }
    std::vector<float> hv(vol.size()), hv2(vol.size());
    CK(hipMemcpy(hv.data(), gvol, vol.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hv2.data(), gvol2, vol.size() * 4, hipMemcpyDeviceToHost));
    double mx = 0, md = 0;
    for (size_t i = 0; i < hv.size(); ++i) { mx = std::max(mx, (double)std::fabs(hv[i])); md = std::max(md, (double)std::fabs(hv[i] - hv2[i])); }
#ifdef AHV_BWD_DU_AMAX
    printf("B=%d N=%ld: head %.3f ms  dW1 %.3f ms  dV %.3f ms (LDS atomics)  total %.3f ms   (grad_vol[0..2] = %g %g %g)\n", B, N, t[0] / iters,
           t[1] / iters, t[2] / iters, (t[0] + t[1] + t[2]) / iters, hv[0], hv[1], hv[2]);
#else
    printf("B=%d N=%ld: head %.3f ms  dW1 %.3f ms   (grad_vol[0..2] = %g %g %g)\n", B, N, t[0] / iters, t[1] / iters, hv2[0], hv2[1], hv2[2]);
#endif
#ifdef AHV_RMW_STAMPS
    {
        unsigned long long hs[64];
        CK(hipMemcpyFromSymbol(hs, HIP_SYMBOL(ahv::g_rmw_stamps), sizeof(hs)));
        static const char* nm[8] = {"loop head", "wait done", "corners", "dX half 0", "wait ready", "scatter 0", "dX half 1", "scatter 1"};
        const double nh = (double)N / (2.0 * gx) * ((B + gy - 1) / gy);   // hypotheses per slot of workgroup (3, 5), roughly
        printf("   workgroup (3, 5): shader-clock cycles per hypothesis and phase (about %.0f hypotheses per slot)\n", nh);
        for (int w = 0; w < 8; ++w) {
            printf("   wave %d (slot %d member %d):", w, w >> 2, w & 3);
            double tot = 0;
            for (int i = 0; i < 8; ++i) { printf(" %s %.0f |", nm[i], hs[w * 8 + i] / nh); tot += hs[w * 8 + i] / nh; }
            printf(" total %.0f\n", tot);
        }
    }
#endif
#ifdef AHV_BWD_DU_AMAX
    printf("           dV read-modify-write kernel %.3f ms  total with it %.3f ms   max |difference| / max |dV| = %.2e\n", t[3] / iters,
           (t[0] + t[1] + t[3]) / iters, md / mx);
#else
    (void)md; (void)mx;
    printf("           dV read-modify-write kernel %.3f ms  recomputing backward %.3f ms\n", t[3] / iters, (t[0] + t[1] + t[3]) / iters);
#endif
    printf("           head kernel of the training pair (u saved by the forward) %.3f ms  total with it %.3f ms\n", t[4] / iters,
           (t[4] + t[1] + t[3]) / iters);
    return 0;
}
