// tile_probe: linear_tile_kernel (ahv_encoder.hip) alone at chosen (M, K, N / H), events around 20 launches.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DAHV_DIAG_TILE=<mask>] -I3dahv_amd/csrc -Iinclude tools/tile_probe.cpp -o tools/tile_probe
// AHV_DIAG_TILE (wrong results, timing only): bit 1 no tile loads, 2 no fragment reads, 3 no barrier.
#include "../3dahv_amd/csrc/ahv_encoder.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void fillk(float* p, size_t n, unsigned seed)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = ((float)(h & 0xffffff) / 8388608.0f - 1.0f) * 0.05f;
    }
}
static void run(const char* name, int M, int K, int N, int KS, int H, int tile = 128)
{
    float *X[2], *W[2], *P[2], *b[2];
    for (int i = 0; i < 2; ++i) {
        CK(hipMalloc(&X[i], (size_t)M * K * 4)); CK(hipMalloc(&W[i], (size_t)N * K * 4));
        CK(hipMalloc(&P[i], (size_t)M * N * 4 * (KS > 1 ? KS : 1))); CK(hipMalloc(&b[i], (size_t)N * 4));
        fillk<<<256, 256>>>(X[i], (size_t)M * K, 1 + i); fillk<<<256, 256>>>(W[i], (size_t)N * K, 3 + i); fillk<<<16, 256>>>(b[i], N, 5 + i);
    }
    ahv::LinSpec sp[2];
    for (int i = 0; i < 2; ++i) sp[i] = ahv::LinSpec{X[i], W[i], P[i], b[i], H > 0 ? N : N};
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ts;
    for (int it = 0; it < 25; ++it) {
        CK(hipEventRecord(e0));
        CK(ahv::launch_linear_tile(sp, 2, K, K, M, K, KS, H, 0, tile));
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (it >= 5) ts.push_back(ms * 1e3f);
    }
    std::sort(ts.begin(), ts.end());
    const double flop = 2.0 * 2 * M * (double)K * N;
    printf("%-28s M %5d K %5d N %5d KS %d: min %7.1f median %7.1f us  %6.1f TFLOP/s = %.3f of 157.3\n", name, M, K, N, KS, ts[0], ts[ts.size() / 2],
           flop / ts[ts.size() / 2] * 1e-6, flop / ts[ts.size() / 2] * 1e-6 / 157.3);
    for (int i = 0; i < 2; ++i) { CK(hipFree(X[i])); CK(hipFree(W[i])); CK(hipFree(P[i])); CK(hipFree(b[i])); }
}
int main()
{
    for (int rep = 0; rep < 2; ++rep) {
        run("FF-in GEGLU", 2048, 512, 4096, 1, 2048);
        run("FF-in GEGLU, K x 4", 2048, 2048, 4096, 1, 2048);
        run("plain, same shape", 2048, 512, 4096, 1, 0);
        run("plain, K x 4", 2048, 2048, 4096, 1, 0);
        run("FF-out split-K 4", 2048, 2048, 256, 4, 0);
        run("FF-out split-K 8", 2048, 2048, 256, 8, 0);
        run("qkv projection shape", 2048, 256, 768, 1, 0);
        run("256 -> 256 shape", 2048, 256, 256, 1, 0);
        run("qkv, 64-tiles", 2048, 256, 768, 1, 0, 64);
        run("256 -> 256, 64-tiles", 2048, 256, 256, 1, 0, 64);
        run("qkv, 64-tiles, M 1024", 1024, 256, 768, 1, 0, 64);
        run("qkv, 64-tiles, M 512", 512, 256, 768, 1, 0, 64);
        run("qkv, 64-tiles, M 256", 256, 256, 768, 1, 0, 64);
        run("FF-out split-K 4, 64-tiles", 2048, 2048, 256, 4, 0, 64);
        run("FF-out split-K 2, 64-tiles", 2048, 2048, 256, 2, 0, 64);
        run("FF-in GEGLU, M / 2", 1024, 512, 4096, 1, 2048);
        run("FF-in GEGLU, M x 2", 4096, 512, 4096, 1, 2048);
    }
    return 0;
}
