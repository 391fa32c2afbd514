// Is v_mfma_f32_16x16x4_f32 the sequential fused-multiply-add chain over its four k terms?
//   D[i][j] = fma(A[i][3], B[3][j], fma(A[i][2], B[2][j], fma(A[i][1], B[1][j], fma(A[i][0], B[0][j], C[i][j]))))
// If yes, two extra output columns of an MFMA GEMM (the attention dot products of the GATConv layers) can be produced by a
// plain v_fma_f32 chain on the VALU with the SAME bits, instead of a 16-column MFMA block of which 14 columns are waste.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/native/bin/mfma_fma_probe tools/native/mfma_fma_probe.hip
// Prints the number of differing outputs for a few operand distributions and K = 256 chains; exit 1 when any differ.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <cstring>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// one wave: C[16][16] = A[16][K] * B[16][K]^T, MFMA chain k ascending; lane (r, q): A operand element t = A[r][4 blk + ... ]
__global__ void mfma_kernel(const float *A, const float *B, float *C, int K)
{
    const int lane = threadIdx.x, r = lane & 15, q = lane >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < K; k0 += 16) {
        const f32x4 av = *reinterpret_cast<const f32x4 *>(A + r * K + k0 + 4 * q);
        const f32x4 bv = *reinterpret_cast<const f32x4 *>(B + r * K + k0 + 4 * q);
        for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t], bv[t], acc, 0, 0, 0);
    }
    for (int reg = 0; reg < 4; ++reg) C[(4 * q + reg) * 16 + r] = acc[reg];
}

// the same sum as ONE fma chain per output in the order the MFMA chain above consumes k:
// for k0: for t: the instruction sums k = k0 + 4 q' + t over q' = 0..3 (the four k of one 16x16x4 are lanes' q)
__global__ void fma_kernel(const float *A, const float *B, float *C, int K, int order)
{
    const int i = threadIdx.x >> 4, j = threadIdx.x & 15;
    float c = 0.f;
    for (int k0 = 0; k0 < K; k0 += 16)
        for (int t = 0; t < 4; ++t) {
            if (order == 0)
                for (int qq = 0; qq < 4; ++qq) { const int k = k0 + 4 * qq + t; c = __builtin_fmaf(A[i * K + k], B[j * K + k], c); }
            else
                for (int qq = 3; qq >= 0; --qq) { const int k = k0 + 4 * qq + t; c = __builtin_fmaf(A[i * K + k], B[j * K + k], c); }
        }
    C[i * 16 + j] = c;
}

int main()
{
    const int K = 256, trials = 400;
    std::vector<float> hA(16 * K), hB(16 * K), c0(256), c1(256), c2(256);
    float *A, *B, *C0, *C1, *C2;
    hipMalloc(&A, 16 * K * 4); hipMalloc(&B, 16 * K * 4); hipMalloc(&C0, 1024); hipMalloc(&C1, 1024); hipMalloc(&C2, 1024);
    long long diff_fwd = 0, diff_rev = 0, total = 0;
    srand(1);
    for (int dist = 0; dist < 4; ++dist) {
        long long d0 = 0, d1 = 0;
        for (int tr = 0; tr < trials; ++tr) {
            for (int n = 0; n < 16 * K; ++n) {
                float u = (float)rand() / RAND_MAX, v = (float)rand() / RAND_MAX;
                if (dist == 0) { hA[n] = u; hB[n] = v - 0.5f; }
                else if (dist == 1) { hA[n] = (u - 0.5f) * expf(20.f * (v - 0.5f)); hB[n] = (v - 0.5f) * expf(20.f * (u - 0.5f)); }
                else if (dist == 2) { hA[n] = u * u * u * u; hB[n] = (v - 0.5f) * 0.125f; }
                else { hA[n] = (rand() % 7 == 0) ? 0.f : (u - 0.5f) * 1e-20f; hB[n] = (v - 0.5f) * 1e-20f; }   // denormal products
            }
            hipMemcpy(A, hA.data(), 16 * K * 4, hipMemcpyHostToDevice);
            hipMemcpy(B, hB.data(), 16 * K * 4, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(mfma_kernel, dim3(1), dim3(64), 0, 0, A, B, C0, K);
            hipLaunchKernelGGL(fma_kernel, dim3(1), dim3(256), 0, 0, A, B, C1, K, 0);
            hipLaunchKernelGGL(fma_kernel, dim3(1), dim3(256), 0, 0, A, B, C2, K, 1);
            hipMemcpy(c0.data(), C0, 1024, hipMemcpyDeviceToHost);
            hipMemcpy(c1.data(), C1, 1024, hipMemcpyDeviceToHost);
            hipMemcpy(c2.data(), C2, 1024, hipMemcpyDeviceToHost);
            for (int n = 0; n < 256; ++n) {
                d0 += memcmp(&c0[n], &c1[n], 4) != 0;
                d1 += memcmp(&c0[n], &c2[n], 4) != 0;
            }
            total += 256;
        }
        printf("distribution %d: mfma vs fma chain (q ascending) %lld / %d differ; (q descending) %lld differ\n", dist, d0, trials * 256, d1);
        diff_fwd += d0; diff_rev += d1;
    }
    printf("TOTAL %lld outputs: forward-order chain differs on %lld, reverse-order on %lld\n", total, diff_fwd, diff_rev);
    return (diff_fwd == 0 || diff_rev == 0) ? 0 : 1;
}
