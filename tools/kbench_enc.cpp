// kbench_enc.cpp -- developer micro-benchmark of forward_2d3d (ahv_encoder.hip) with no Python and no profiler:
// random weights, the forward captured in a hipGraph and replayed, and -- with the launch budget of AHV_ENC_PROBE --
// the marginal cost of every launch (time of the first k launches minus time of the first k - 1).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DAHV_ENC_PROBE -I3dahv_amd/csrc -Iinclude tools/kbench_enc.cpp -o tools/kbench_enc.bin
// Usage: kbench_enc.bin [B] [--each]
#include "../3dahv_amd/csrc/ahv_encoder.hip"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <algorithm>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void fill_kernel(float* p, size_t n, unsigned seed, float scale)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u ^ seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        p[i] = ((float)(h & 0xffffff) / 8388608.0f - 1.0f) * scale;
    }
}

static float* g_pool;
static size_t g_used, g_cap;
static unsigned g_seed = 1;
static const float* param(size_t n, float scale, float offset = 0.0f)
{
    float* p = g_pool + g_used;
    g_used += (n + 63) & ~(size_t)63;
    if (g_used > g_cap) { printf("pool too small\n"); exit(1); }
    fill_kernel<<<256, 256>>>(p, n, g_seed++ * 7919u, scale);
    (void)offset;
    return p;
}

static double replay_us(hipStream_t s, int B, const ahv_aligner_weights& w, const float* l4s, const float* l4t, float* ws,
                        float* vs, float* vt, int budget, int reps, int* launches)
{
    const char* what = "";
    g_enc_probe_budget = budget;
    g_enc_probe_count = 0;
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    const int rc = ahv::forward_2d3d(&w, l4s, l4t, B, ws, vs, vt, s, &what);
    CK(hipStreamEndCapture(s, &g));
    if (rc) { printf("forward_2d3d failed at %s (%d)\n", what, rc); exit(1); }
    if (launches) *launches = g_enc_probe_count;
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 10; ++i) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    double best = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms * 1e3 / reps < best) best = ms * 1e3 / reps;
    }
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return best;
}

int main(int argc, char** argv)
{
    int B = 1; bool each = false; bool stamps = false;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--each")) each = true;
        else if (!strcmp(argv[i], "--stamps")) stamps = true;
        else if (!strcmp(argv[i], "--dup")) g_enc_probe_dup = 1;  // weight-reading launches twice: cold / hot marginal costs
        else B = atoi(argv[i]);
    }
    hipStream_t s; CK(hipStreamCreate(&s));
    g_cap = (size_t)56 << 20;  // floats: 47.9 M parameters + slack
    CK(hipMalloc(&g_pool, g_cap * 4));
    const int depth = 4;
    std::vector<ahv_block_weights> blocks(4 * depth);
    for (auto& b : blocks) {
        b.w_qkv = param(768 * 256, 0.06f); b.w_out = param(256 * 256, 0.06f); b.b_out = param(256, 0.05f);
        b.ln1_g = param(256, 0.1f); b.ln1_b = param(256, 0.1f);
        b.w_ff1 = param(4096 * 512, 0.04f); b.b_ff1 = param(4096, 0.04f);
        b.w_ff2 = param(256 * 2048, 0.02f); b.b_ff2 = param(256, 0.02f);
        b.ln2_g = param(256, 0.1f); b.ln2_b = param(256, 0.1f);
    }
    ahv_aligner_weights w;
    w.w_emb = param(256 * 768, 0.03f); w.w_conv1 = param(256 * 2304, 0.02f); w.w_conv2 = param(256 * 2304, 0.02f);
    w.posemb = param(64 * 256, 1.0f); w.gn_g = param(256, 1.0f); w.gn_b = param(256, 0.1f);
    for (int i = 0; i < 2; ++i) {
        w.w_in[i] = param(256 * 256, 0.06f); w.b_in[i] = param(256, 0.05f);
        w.w_out[i] = param(256 * 256, 0.06f); w.b_out[i] = param(256, 0.05f);
    }
    w.w3d_1 = param(32 * 1024, 0.03f); w.w3d_2 = param(16 * 512, 0.04f);
    w.blocks = blocks.data(); w.depth = depth;
    const size_t nin = (size_t)B * 768 * 64, nvol = (size_t)B * 16 * 512;
    float *l4s, *l4t, *vs, *vt, *ws;
    CK(hipMalloc(&l4s, nin * 4)); CK(hipMalloc(&l4t, nin * 4)); CK(hipMalloc(&vs, nvol * 4)); CK(hipMalloc(&vt, nvol * 4));
    fill_kernel<<<256, 256>>>(l4s, nin, 11u, 1.0f);
    fill_kernel<<<256, 256>>>(l4t, nin, 13u, 1.0f);
    const size_t wsf = ahv::forward_2d3d_workspace_floats(B);
    CK(hipMalloc(&ws, wsf * 4));
    CK(hipDeviceSynchronize());
    int launches = 0;
    const double all = replay_us(s, B, w, l4s, l4t, ws, vs, vt, 1 << 30, 200, &launches);
    std::vector<float> h(nvol);
    CK(hipMemcpy(h.data(), vs, nvol * 4, hipMemcpyDeviceToHost));
    double cs = 0; for (float v : h) cs += v;
    printf("{\"bench\": \"forward_2d3d\", \"B\": %d, \"launches\": %d, \"hipgraph_us\": %.2f, \"checksum\": %.6f}\n", B, launches, all, cs);
    if (stamps) {  // in-kernel timeline of every linear launch (wave 0 of each workgroup), microseconds after the first entry
        unsigned long long* d;
        const size_t n = (size_t)256 * 8192;
        CK(hipMalloc(&d, n * 8));
        CK(hipMemset(d, 0, n * 8));
        g_enc_probe_stamps = d;
        replay_us(s, B, w, l4s, l4t, ws, vs, vt, 1 << 30, 20, nullptr);
        g_enc_probe_stamps = nullptr;
        std::vector<unsigned long long> hs(n);
        CK(hipMemcpy(hs.data(), d, n * 8, hipMemcpyDeviceToHost));
        static const char* slot_name[7] = {"entry", "prologue operands", "tile written", "tile barrier", "mfma done", "reduce barrier", "exit"};
        for (int k = 0; k < launches; ++k) {
            const unsigned long long* st = hs.data() + (size_t)k * 8192;
            unsigned long long t0 = ~0ull; int nwg = 0;
            for (int g = 0; g < 1024; ++g) if (st[g * 8]) { if (st[g * 8] < t0) t0 = st[g * 8]; nwg = g + 1; }
            if (!nwg) continue;
            printf("launch %2d %-48s %d workgroups\n", k + 1, g_enc_probe_names[k], nwg);
            if (strstr(g_enc_probe_names[k], "linear_tile_kernel")) {   // slots: 0 entry, 1 first tiles landed, 4 K loop done, 6 exit
                std::vector<double> ent, ex;
                double pro = 0, kl = 0, epi = 0; int c = 0;
                for (int g = 0; g < nwg; ++g) if (st[g * 8] && st[g * 8 + 6]) {
                    ent.push_back((st[g * 8] - t0) * 0.01); ex.push_back((st[g * 8 + 6] - t0) * 0.01);
                    pro += (st[g * 8 + 1] - st[g * 8]) * 0.01; kl += (st[g * 8 + 4] - st[g * 8 + 1]) * 0.01; epi += (st[g * 8 + 6] - st[g * 8 + 4]) * 0.01; ++c;
                }
                if (!c) continue;
                std::sort(ent.begin(), ent.end()); std::sort(ex.begin(), ex.end());
                printf("    entry time deciles:");
                for (int d = 0; d <= 10; ++d) printf(" %.2f", ent[std::min(ent.size() - 1, d * ent.size() / 10)]);
                printf("\n    exit time deciles: ");
                for (int d = 0; d <= 10; ++d) printf(" %.2f", ex[std::min(ex.size() - 1, d * ex.size() / 10)]);
                printf("\n    per workgroup: first tiles landed +%.2f us, K loop %.2f us, epilogue %.2f us\n", pro / c, kl / c, epi / c);
                continue;
            }
            if (strstr(g_enc_probe_names[k], "attention")) {   // slots: entry, scores, softmax share, merge barrier, O share, projection, exit
                std::vector<double> ent, ex;
                for (int g = 0; g < nwg; ++g) if (st[g * 8]) { ent.push_back((st[g * 8] - t0) * 0.01); ex.push_back((st[g * 8 + 6] - t0) * 0.01); }
                std::sort(ent.begin(), ent.end()); std::sort(ex.begin(), ex.end());
                printf("    entry time deciles:");
                for (int d = 0; d <= 10; ++d) printf(" %.2f", ent[std::min(ent.size() - 1, d * ent.size() / 10)]);
                printf("\n    exit time deciles: ");
                for (int d = 0; d <= 10; ++d) printf(" %.2f", ex[std::min(ex.size() - 1, d * ex.size() / 10)]);
                printf("\n");
                static const char* an_tile[7] = {"entry", "scores done", "softmax share", "merge barrier", "O share done", "projection done", "exit"};
                static const char* an_pair[7] = {"entry", "scores done", "softmax done", "P V done", "W_out in LDS", "-", "exit"};
                const char* const* an = strstr(g_enc_probe_names[k], "sample_head") ? an_pair : an_tile;
                double prevs = 0;
                for (int sl = 1; sl < 7; ++sl) {
                    double sum = 0; int c = 0;
                    for (int g = 0; g < nwg; ++g) if (st[g * 8] && st[g * 8 + sl]) { sum += (double)(st[g * 8 + sl] - st[g * 8]) * 0.01; ++c; }
                    if (c) printf("    %-16s +%6.2f us after entry (phase %5.2f)\n", an[sl], sum / c, sum / c - prevs);
                    if (c) prevs = sum / c;
                }
                continue;
            }
            for (int sl = 0; sl < 7; ++sl) {
                double mn = 1e30, mx = 0, sum = 0; int c = 0;
                for (int g = 0; g < nwg; ++g) {
                    const unsigned long long v = st[g * 8 + sl];
                    if (!v) continue;
                    const double us = (double)(v - t0) * 0.01;
                    if (us < mn) mn = us;
                    if (us > mx) mx = us;
                    sum += us; ++c;
                }
                if (c) printf("    %-18s min %6.2f  avg %6.2f  max %6.2f us\n", slot_name[sl], mn, sum / c, mx);
            }
        }
    }
    if (each) {
        std::vector<const char*> names(g_enc_probe_names, g_enc_probe_names + launches);
        double prev = 0;
        std::map<std::string, std::pair<int, double>> agg;
        for (int k = 1; k <= launches; ++k) {
            const double t = replay_us(s, B, w, l4s, l4t, ws, vs, vt, k, 200, nullptr);
            printf("  %2d %-52s +%.2f us (cumulative %.2f)\n", k, names[k - 1], t - prev, t);
            auto& a = agg[names[k - 1]]; a.first++; a.second += t - prev;
            prev = t;
        }
        for (auto& kv : agg) printf("{\"kernel\": \"%s\", \"launches\": %d, \"total_us\": %.2f, \"avg_us\": %.2f}\n", kv.first.c_str(), kv.second.first, kv.second.second, kv.second.second / kv.second.first);
        if (g_enc_probe_dup) {  // what a perfect weight prefetch could save: sum over the weight-reading launches of (cold - hot)
            double cold = 0, hot = 0;
            for (auto& kv : agg) {
                const std::string again = kv.first + " [again: weights hot]";
                if (agg.count(again)) { cold += kv.second.second; hot += agg[again].second; }
            }
            printf("{\"B\": %d, \"weight_reading_launches_cold_us\": %.2f, \"same_launches_weights_hot_us\": %.2f, \"perfect_prefetch_saves_us\": %.2f}\n", B, cold, hot, cold - hot);
        }
    }
    return 0;
}
