// kbench.cpp -- developer micro-benchmark for the fused scorer (not part of the product).
// Includes the kernel source directly so that diagnostic builds (-DAHV_STAMPS) can read the
// in-kernel cycle stamps.  Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DAHV_STAMPS] \
//                                   -I3dahv_amd/csrc tools/kbench.cpp -o tools/kbench
// Run (on the GPU box):  tools/kbench [N] [iters] [variant] [spare_cus] [no_teams]
//   variant 3 = fp32 kernel (target features given), 4 = split-f16, 5 = fp32 with the target features built in the
//   launch (ahv_verify_pair_f32); without [variant]: 3, 4 and 5 in turn.
#include "../3dahv_amd/csrc/ahv_score.hip"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

int main(int argc, char** argv)
{
    const long N = argc > 1 ? atol(argv[1]) : 50000;
    const int iters = argc > 2 ? atoi(argv[2]) : 20;
    std::mt19937 rng(0);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> vol(8192), ft(2048), R(N * 9), W1(32 * 384), W2(1024), b2(32), vtgt(8192);
    for (auto& x : vtgt) x = 1.15f * nd(rng);
    for (auto& x : vol) x = 1.15f * nd(rng);
    for (auto& x : ft) x = nd(rng) / 5.6f;
    for (auto& x : W1) x = nd(rng) * 0.03f;
    for (auto& x : W2) x = nd(rng) * 0.1f;
    for (auto& x : b2) x = nd(rng) * 0.1f;
    for (long n = 0; n < N; ++n) {  // Haar rotations from normalised Gaussian quaternions
        double q[4], s = 0;
        for (double& c : q) { c = nd(rng); s += c * c; }
        const double t = 2.0 / s, r = q[0], i = q[1], j = q[2], k = q[3];
        const double m[9] = {1 - t * (j * j + k * k), t * (i * j - k * r), t * (i * k + j * r),
                             t * (i * j + k * r), 1 - t * (i * i + k * k), t * (j * k - i * r),
                             t * (i * k - j * r), t * (j * k + i * r), 1 - t * (i * i + j * j)};
        for (int e = 0; e < 9; ++e) R[n * 9 + e] = (float)m[e];
    }
    float *dvol, *dft, *dR, *dW1, *dW2, *db2, *dsc, *dvt;
    int64_t* dkey;
    const int spare = argc > 4 ? atoi(argv[4]) : 0;
    const bool no_teams = argc > 5 && atoi(argv[5]) != 0;
    CK(hipMalloc(&dvol, vol.size() * 4)); CK(hipMalloc(&dft, ft.size() * 4)); CK(hipMalloc(&dR, R.size() * 4));
    CK(hipMalloc(&dW1, W1.size() * 4)); CK(hipMalloc(&dW2, W2.size() * 4)); CK(hipMalloc(&db2, b2.size() * 4));
    CK(hipMalloc(&dsc, N * 4)); CK(hipMalloc(&dkey, 8)); CK(hipMalloc(&dvt, vtgt.size() * 4));
    CK(hipMemcpy(dvt, vtgt.data(), vtgt.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dvol, vol.data(), vol.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dft, ft.data(), ft.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dR, R.data(), R.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dW1, W1.data(), W1.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dW2, W2.data(), W2.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db2, b2.data(), b2.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dkey, 0, 8));
    int cu = 0;
    CK(hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, 0));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ref(N);
    {   // the target features of variant 3 / 4 = what variant 5 computes in the launch (so that the three can be compared)
        ahv::ScoreLaunch a = {dvol, dvt, true, dR, 0, 0, dW1, dW2, db2, 1, 0, nullptr, nullptr, dft, cu, 0, false, false, nullptr};
        CK(ahv::launch_score_hypotheses(a, 0));
        CK(hipDeviceSynchronize());
        const ahv::ScorePlan p = ahv::plan_score_launch(1, N, cu, spare, !no_teams);
        printf("N %ld: grid %d x %d, %ld hypotheses by single waves, %ld by teams (spare CUs %d)\n", N, p.gx, p.gy,
               (long)p.n_main, (long)(N - p.n_main), spare);
    }
    for (int variant = (argc > 3 ? atoi(argv[3]) : 3); variant < (argc > 3 ? atoi(argv[3]) + 1 : 6); ++variant) {
        // 3 = the product's fp32 kernel, 4 = its opt-in split-f16 sibling, 5 = fp32 + in-launch target features (0-2,
        // the retired round-1 formulations, left the tree in round 3; the numbering is kept so that old logs stay comparable)
        if (variant < 3) { printf("variants 0-2 are retired\n"); return 1; }
        auto launch = [&]() {
            ahv::ScoreLaunch a = {dvol, variant == 5 ? dvt : dft, variant == 5, dR, 0, 0, dW1, dW2, db2, 1, N, dsc, dkey, nullptr,
                                  cu, spare, variant == 4, no_teams, nullptr};
            return ahv::launch_score_hypotheses(a, 0);
        };
        for (int rep = 0; rep < 3; ++rep) {
            for (int w = 0; w < 3; ++w)
                CK(launch());
            CK(hipEventRecord(e0, 0));
            for (int it = 0; it < iters; ++it)
                CK(launch());
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            ms /= iters;
            printf("variant %d: %.4f ms  %.3e hyp/s  %.1f TFLOP/s (%.1f%% of 157.3)\n", variant, ms, N / (ms * 1e-3),
                   N / (ms * 1e-3) * 1839104 / 1e12, N / (ms * 1e-3) * 1839104 / 1e12 / 157.3 * 100);
        }
        std::vector<float> sc(N);
        CK(hipMemcpy(sc.data(), dsc, N * 4, hipMemcpyDeviceToHost));
        if (variant == 3) ref = sc;
        double md = 0;
        for (long n = 0; n < N; ++n) md = std::max(md, (double)std::abs(sc[n] - ref[n]));
        printf("variant %d: max |score - variant3| = %.3g, score[0]=%.6f\n", variant, md, sc[0]);
    }
#ifdef AHV_STAMPS
    // stamps belong to the last variant run (3 = dual): 4 x (gather, gemm1) quarters, gemm2, score+tail
    std::vector<unsigned long long> st(2048 * 16);
    CK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(ahv::g_stamps), st.size() * 8));
    double sum[11] = {0};
    for (int w = 0; w < 2048; ++w) for (int i = 0; i < 11; ++i) sum[i] += (double)st[w * 16 + i];
    const char* names[10] = {"gather0", "gemm0", "gather1", "gemm1", "gather2", "gemm2q", "gather3", "gemm3", "gemm2", "score+tail"};
    double tot = 0;
    for (int i = 0; i < 10; ++i) tot += sum[i] / sum[10];
    for (int i = 0; i < 10; ++i) printf("  %-10s %8.0f ticks/hyp (%.1f%%)\n", names[i], sum[i] / sum[10], 100 * sum[i] / sum[10] / tot);
    printf("  total      %8.0f ticks/hyp per wave (two waves share a SIMD), rounds/wave %.1f\n", tot, sum[10] / 2048);
    // per-workgroup real-time stamps (100 MHz) of the last launch: prologue length, loop length, finish spread per XCC
    std::vector<unsigned long long> wg(1024 * 4);
    CK(hipMemcpyFromSymbol(wg.data(), HIP_SYMBOL(ahv::g_wgstamps), wg.size() * 8));
    unsigned long long t0 = ~0ull, t1 = 0;
    int nwg = 0;
    for (int w = 0; w < 1024; ++w) if (wg[4 * w + 2]) { t0 = std::min(t0, wg[4 * w]); t1 = std::max(t1, wg[4 * w + 2]); ++nwg; }
    double pro = 0, loop = 0, idle = 0, entry = 0;
    double xs[16] = {0}, xe[16] = {0}; int xn[16] = {0};
    for (int w = 0; w < 1024; ++w) if (wg[4 * w + 2]) {
        pro += (wg[4 * w + 1] - wg[4 * w]) * 0.01; loop += (wg[4 * w + 2] - wg[4 * w + 1]) * 0.01;
        idle += (t1 - wg[4 * w + 2]) * 0.01; entry += (wg[4 * w] - t0) * 0.01;
        const int x = (int)(wg[4 * w + 3] & 15); xs[x] += (wg[4 * w + 2] - wg[4 * w + 1]) * 0.01; xe[x] += (t1 - wg[4 * w + 2]) * 0.01; ++xn[x];
    }
    printf("  workgroups %d: first entry -> last exit %.1f us; mean entry delay %.1f us, prologue %.1f us, loop %.1f us, idle at the end %.1f us\n",
           nwg, (t1 - t0) * 0.01, entry / nwg, pro / nwg, loop / nwg, idle / nwg);
    {   // distribution of the workgroups' exit times (relative to the first entry) and of their loop lengths
        std::vector<double> ex, lp;
        for (int w = 0; w < 1024; ++w) if (wg[4 * w + 2]) { ex.push_back((wg[4 * w + 2] - t0) * 0.01); lp.push_back((wg[4 * w + 2] - wg[4 * w + 1]) * 0.01); }
        std::sort(ex.begin(), ex.end()); std::sort(lp.begin(), lp.end());
        auto q = [](const std::vector<double>& v, double f) { return v[(size_t)(f * (v.size() - 1))]; };
        printf("  exit time   min %.1f  p10 %.1f  p50 %.1f  p90 %.1f  p99 %.1f  max %.1f us\n", ex.front(), q(ex, .1), q(ex, .5), q(ex, .9), q(ex, .99), ex.back());
        printf("  loop length min %.1f  p10 %.1f  p50 %.1f  p90 %.1f  p99 %.1f  max %.1f us\n", lp.front(), q(lp, .1), q(lp, .5), q(lp, .9), q(lp, .99), lp.back());
        // the ten slowest workgroups: index (= blockIdx.x), XCC, loop length
        std::vector<std::pair<double, int>> byexit;
        for (int w = 0; w < 1024; ++w) if (wg[4 * w + 2]) byexit.push_back({(wg[4 * w + 2] - t0) * 0.01, w});
        std::sort(byexit.begin(), byexit.end());
        printf("  slowest:");
        for (size_t i = byexit.size() >= 10 ? byexit.size() - 10 : 0; i < byexit.size(); ++i)
            printf(" wg%d(xcc%d %.1f)", byexit[i].second, (int)(wg[4 * byexit[i].second + 3] & 15), byexit[i].first);
        printf("\n");
    }
    {   // shader clock held during the loop: s_memtime ticks per 10-ns real-time tick, median over workgroups
        std::vector<unsigned long long> ck(1024 * 2);
        CK(hipMemcpyFromSymbol(ck.data(), HIP_SYMBOL(ahv::g_wgclk), ck.size() * 8));
        std::vector<double> ghz;
        for (int w = 0; w < 1024; ++w) if (wg[4 * w + 2] > wg[4 * w + 1]) ghz.push_back((double)(ck[2 * w + 1] - ck[2 * w]) / (double)(wg[4 * w + 2] - wg[4 * w + 1]) * 0.1);
        std::sort(ghz.begin(), ghz.end());
        if (!ghz.empty()) printf("  shader clock during the loop (s_memtime / s_memrealtime): median %.3f GHz, min %.3f, max %.3f\n", ghz[ghz.size() / 2], ghz.front(), ghz.back());
    }
    for (int x = 0; x < 16; ++x) if (xn[x]) printf("    XCC %d: %d workgroups, mean loop %.1f us, mean idle at the end %.1f us\n", x, xn[x], xs[x] / xn[x], xe[x] / xn[x]);
#endif
    return 0;
}
