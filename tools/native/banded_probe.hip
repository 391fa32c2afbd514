// Stand-alone timing of gat_layer_banded_kernel (csrc/nsc_gat_banded.h) on the GATConv layer of the reference's shape
// (hidden 256, temporal chain with 5 neighbours, edge_dim 2) at M nodes:
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -I include -o tools/native/bin/banded_probe tools/native/banded_probe.hip
//   banded_probe [M=4541] [reps=300]
// Every (ACC, NST) configuration: time per launch from HIP events over back-to-back launches, and the output compared bit for
// bit with the first configuration's (parity against the generic kernels is tests/test_gat_gpu.py's job).
// Diagnostic builds: -DNSC_BAND_CLOCK (in-kernel s_memtime stamps: where a workgroup's time goes), -DNSC_BAND_ABL=1|2|4
// (no aggregation / no attention chains / no entry prefetch: timings only, outputs differ by design).
#include "../../neural-spectral-codec_amd/csrc/nsc_gat.hip"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

static float *dalloc(size_t n, unsigned seed, float scale, float offset = 0.f)
{
    std::vector<float> h(n);
    unsigned s = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < n; ++i) {
        s = s * 1664525u + 1013904223u;
        h[i] = ((float)(s >> 8) / 16777216.0f - 0.5f) * scale + offset;
    }
    float *d = nullptr;
    if (hipMalloc(&d, n * sizeof(float)) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); exit(2); }
    hipMemcpy(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice);
    return d;
}

template <typename F>
static float time_us(F launch, int reps)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 10; ++i) launch();
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return ms * 1000.0f / reps;
}

static std::vector<float> g_ref;
static int g_bad = 0;

template <int ACC, int NST>
static void run_cfg(BandArgs a, int reps)
{
    constexpr unsigned lds = NST * (16 * ACC + 64) * 256;
    if (lds > 160 * 1024) return;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&gat_layer_banded_kernel<ACC, NST>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        printf("  ACC %d NST %d: LDS opt-in refused\n", ACC, NST);
        return;
    }
    constexpr int own = 16 * ACC - 2 * NSC_BAND_HALO;
    const dim3 grid(a.H / 64, (a.M + own - 1) / own);
    hipMemset(a.out, 0xff, (size_t)a.M * a.H * 4);
#ifdef NSC_BAND_CLOCK
    const size_t tiles = (size_t)grid.x * grid.y;
    hipMalloc(&a.clk, tiles * 8 * sizeof(unsigned long long));
    hipMemset(a.clk, 0, tiles * 8 * sizeof(unsigned long long));
#endif
    auto launch = [&]() { hipLaunchKernelGGL((gat_layer_banded_kernel<ACC, NST>), grid, dim3(640), lds, 0, a, a.Bx); };
    const float us = time_us(launch, reps);
    std::vector<float> out((size_t)a.M * a.H);
    hipMemcpy(out.data(), a.out, out.size() * 4, hipMemcpyDeviceToHost);
    size_t diff = 0;
    if (g_ref.empty()) g_ref = out;
    else for (size_t i = 0; i < out.size(); ++i) diff += memcmp(&out[i], &g_ref[i], 4) != 0;
    if (diff && !NSC_BAND_ABL) g_bad = 1;
    printf("  ACC %d (tile %3d rows, %3d owned) NST %d  grid %4u  lds %6u B : %7.2f us   %s\n", ACC, 16 * ACC, own, NST,
           grid.x * grid.y, lds, us, diff ? "DIFFERS from the first configuration" : "same bits");
#ifdef NSC_BAND_CLOCK
    std::vector<unsigned long long> c(tiles * 8);
    hipMemcpy(c.data(), a.clk, c.size() * 8, hipMemcpyDeviceToHost);
    double ph[7] = {0, 0, 0, 0, 0, 0, 0};
    for (size_t t = 0; t < tiles; ++t) {
        const unsigned long long *s = &c[t * 8];
        ph[0] += (double)(s[1] - s[0]);      // entry -> first chunk landed (barrier 0)
        ph[1] += (double)(s[2] - s[1]);      // main loop
        ph[2] += (double)(s[3] - s[2]);      // two barriers + tile / entries to LDS
        ph[3] += (double)(s[4] - s[3]);      // softmax + aggregation + stores issued
        ph[4] += (double)(s[5] - s[0]);      // staging: entry -> prologue chunks issued
        ph[5] += (double)(s[6] - s[0]);      // staging: entry -> last chunk landed
    }
    const char *nm[6] = {"entry -> barrier 0 (first chunk landed)", "main loop", "barriers + tile / entries to LDS",
                         "softmax + aggregation + stores issued", "staging: entry -> prologue issued",
                         "staging: entry -> last chunk landed"};
    for (int i = 0; i < 6; ++i) printf("      %-44s %8.0f cycles = %5.2f us at 2.4 GHz\n", nm[i], ph[i] / tiles, ph[i] / tiles / 2400.0);
    hipFree(a.clk);
#endif
}

int main(int argc, char **argv)
{
    const int M = argc > 1 ? atoi(argv[1]) : 4541, reps = argc > 2 ? atoi(argv[2]) : 300, H = 256;
    BandArgs a = {};
    a.A = dalloc((size_t)M * H, 1, 2.0f);
    a.B = dalloc((size_t)H * H, 2, 0.125f);
    a.Bx = dalloc(2 * (size_t)H + 8, 3, 0.25f);
    a.M = M; a.H = H;
    a.v = a.Bx + 2 * H;
    a.bias = dalloc(H, 4, 0.2f);
    a.bn_w = dalloc(H, 5, 0.5f, 1.0f); a.bn_b = dalloc(H, 6, 0.2f); a.bn_mean = dalloc(H, 7, 0.2f); a.bn_var = dalloc(H, 8, 0.5f, 1.0f);
    a.bn_eps = 1e-5f; a.slope = 0.2f; a.relu = 1;
    a.resid = a.A;
    hipMalloc(&a.out, (size_t)M * H * 4);
    // banded entries of the temporal chain (offsets -2, -1, 1, 2 in that order, self loop last), random edge attributes
    std::vector<f32x4> ent((size_t)M * 8);
    unsigned s = 99;
    int eidx = 0;
    for (int i = 0; i < M; ++i) {
        int slot = 0;
        const int offs[5] = {-2, -1, 1, 2, 0};
        for (int o = 0; o < 5; ++o) {
            const int j = i + offs[o];
            if (j < 0 || j >= M) continue;
            s = s * 1664525u + 1013904223u; const float e0 = (float)(s >> 8) / 16777216.0f;
            s = s * 1664525u + 1013904223u; const float e1 = (float)(s >> 8) / 16777216.0f;
            f32x4 e; e.x = __builtin_bit_cast(float, j); e.y = e0; e.z = e1; e.w = __builtin_bit_cast(float, eidx++);
            ent[(size_t)i * 8 + slot++] = e;
        }
        for (; slot < 8; ++slot) { f32x4 e; e.x = __builtin_bit_cast(float, i); e.y = 0; e.z = 0; e.w = __builtin_bit_cast(float, -1); ent[(size_t)i * 8 + slot] = e; }
    }
    f32x4 *dent; hipMalloc(&dent, ent.size() * 16); hipMemcpy(dent, ent.data(), ent.size() * 16, hipMemcpyHostToDevice);
    a.ent = dent;
    printf("gat_layer_banded_kernel, M = %d, H = %d (lin GEMM %.2f GFLOP = %.2f us at the 157.3 TF f32-MFMA peak), picked ACC = %d\n", M, H,
           2.0 * M * H * H / 1e9, 2.0 * M * H * H / 157.3e6, band_pick_acc(M, H));
    run_cfg<5, 3>(a, reps); run_cfg<5, 4>(a, reps);
    run_cfg<1, 3>(a, reps); run_cfg<1, 4>(a, reps);
    run_cfg<2, 3>(a, reps); run_cfg<2, 4>(a, reps);
    run_cfg<3, 3>(a, reps); run_cfg<3, 4>(a, reps);
    run_cfg<4, 3>(a, reps); run_cfg<4, 4>(a, reps);
    run_cfg<6, 3>(a, reps); run_cfg<6, 4>(a, reps);
    return g_bad;
}
