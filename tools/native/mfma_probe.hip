// Stand-alone probe (hipcc --offload-arch=gfx950 -O3 -o mfma_probe mfma_probe.hip): what sustains v_mfma_f32_16x16x4_f32 in a
// loop shaped like the GEMM kernels of csrc/nsc_gat.hip?  Grid of 256-thread workgroups, each wave runs ITER chunks of
// 32 MFMAs (= one 64-deep chunk of a 32 x 16 wave tile) on CHAINS accumulators, with optional per-chunk extras:
//   mode bit 1: 12 ds_read_b128 operand reads per chunk (3 per 8 MFMAs)      bit 2: 6 ds_write_b128 per chunk
//   mode bit 4: __syncthreads per chunk                                        bit 8: operands loaded from LDS feed the MFMAs
// Prints TFLOP/s for WGS workgroups per CU (occupancy) -- compare with the 157.3 TF peak.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CHAINS, int MODE>
__global__ __launch_bounds__(256) void probe(float *out, int iters)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    f32x4 acc[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 a = {1.0f + lane, 0.5f, 0.25f, 2.0f}, b = {0.5f, 1.5f + wave, 0.125f, 1.0f};
    for (int i = tid; i < 96 * 68; i += 256) lds[i] = (float)(i & 7);
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
        if (MODE & 2) {
#pragma unroll
            for (int w = 0; w < 6; ++w) *reinterpret_cast<f32x4 *>(&lds[((tid + 256 * w) >> 4) * 68 + 4 * (tid & 15)]) = a;
        }
        if (MODE & 4) __syncthreads();
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            f32x4 av0 = a, av1 = a, bv = b;
            if (MODE & 1) {
                const f32x4 x0 = *reinterpret_cast<const f32x4 *>(&lds[(r) * 68 + 16 * d + 4 * q]);
                const f32x4 x1 = *reinterpret_cast<const f32x4 *>(&lds[(16 + r) * 68 + 16 * d + 4 * q]);
                const f32x4 x2 = *reinterpret_cast<const f32x4 *>(&lds[(32 + wave * 16 + r) * 68 + 16 * d + 4 * q]);
                if (MODE & 8) { av0 = x0; av1 = x1; bv = x2; }
                else { a.x += x0.x * 1e-30f + x1.y * 1e-30f + x2.z * 1e-30f; }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                acc[(2 * t) % CHAINS] = __builtin_amdgcn_mfma_f32_16x16x4f32(av0[t], bv[t], acc[(2 * t) % CHAINS], 0, 0, 0);
                acc[(2 * t + 1) % CHAINS] = __builtin_amdgcn_mfma_f32_16x16x4f32(av1[t], bv[t], acc[(2 * t + 1) % CHAINS], 0, 0, 0);
            }
        }
    }
    f32x4 s = acc[0];
#pragma unroll
    for (int c = 1; c < CHAINS; ++c) s += acc[c];
    out[(size_t)blockIdx.x * 256 + tid] = s.x + s.y + s.z + s.w;
}

template <int CHAINS, int MODE>
static void run(const char *name, float *out, int wgs_per_cu, int lds_bytes)
{
    const int iters = 400, grid = 256 * wgs_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<CHAINS, MODE>), dim3(grid), dim3(256), lds_bytes, 0, out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<CHAINS, MODE>), dim3(grid), dim3(256), lds_bytes, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)grid * 4 * iters * 32 * 2048.0;
    printf("%-44s chains %d  wg/CU %d  %.3f ms  %.1f TFLOP/s (%.0f %%)\n", name, CHAINS, wgs_per_cu, ms, flop / ms / 1e9, flop / ms / 1e9 / 157.3 * 100);
}

int main()
{
    float *out;
    hipMalloc(&out, (size_t)256 * 8 * 256 * sizeof(float));
    const int L = 96 * 68 * 4;          // 26 KB: up to 6 workgroups per CU
    for (int w : {1, 2, 3, 4}) {
        if (w == 1) { run<2, 0>("MFMA only", out, 1, L); run<4, 0>("MFMA only", out, 1, L); run<8, 0>("MFMA only", out, 1, L); }
        if (w == 2) { run<2, 0>("MFMA only", out, 2, L); run<4, 0>("MFMA only", out, 2, L); }
        if (w == 3) run<2, 0>("MFMA only", out, 3, L);
        if (w == 4) run<2, 0>("MFMA only", out, 4, L);
    }
    for (int w : {2, 3}) {
        if (w == 2) {
            run<2, 4>("+ barrier per chunk", out, 2, L);
            run<2, 1>("+ operand reads (unused)", out, 2, L);
            run<2, 9>("+ operand reads feeding the MFMAs", out, 2, L);
            run<2, 13>("+ reads feeding MFMAs + barrier", out, 2, L);
            run<2, 15>("+ reads + staging writes + barrier", out, 2, L);
            run<4, 15>("+ reads + staging writes + barrier", out, 2, L);
        } else {
            run<2, 4>("+ barrier per chunk", out, 3, L);
            run<2, 9>("+ operand reads feeding the MFMAs", out, 3, L);
            run<2, 15>("+ reads + staging writes + barrier", out, 3, L);
        }
    }
    hipFree(out);
    return 0;
}
