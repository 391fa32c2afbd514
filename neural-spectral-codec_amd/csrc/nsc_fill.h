// Fills as KERNELS.  Every entry point of include/nsc.h may be captured into a hipGraph, and a hipMemsetAsync captured into a graph
// becomes a memset NODE: on ROCm 7.2 such a node was observed (round 4, the captured training step) not to be ordered before the
// kernel node that follows it in the captured stream -- the buffer was accumulated into before it was zeroed.  Kernel nodes keep
// their stream order, so nothing in this library uses hipMemsetAsync / hipMemcpyAsync on a path that can be captured.
#pragma once

__global__ __launch_bounds__(256) void nsc_fill_u32_kernel(unsigned *__restrict__ p, unsigned v, long long n)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long stride = (long long)gridDim.x * 256;
    for (long long j = i; j < n; j += stride) p[j] = v;
}

// n 32-bit words of `p` := v, on `st`
inline bool nsc_fill_u32(hipStream_t st, void *p, unsigned v, long long n)
{
    if (n <= 0) return true;
    long long b = (n + 255) / 256;
    if (b > 4096) b = 4096;
    hipLaunchKernelGGL(nsc_fill_u32_kernel, dim3((unsigned)b), dim3(256), 0, st, static_cast<unsigned *>(p), v, n);
    return true;
}
