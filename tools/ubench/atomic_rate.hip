// atomic_rate: returning atomicAdd throughput on K words spread over K * stride bytes, every wave of a full grid claiming in a loop
// (the pattern of k_trace's claim cursor).  hipcc --offload-arch=gfx950 -O3 tools/ubench/atomic_rate.hip -o tools/ubench/atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void k(unsigned *ctr, unsigned k_words, unsigned stride_words, unsigned iters, unsigned *sink)
{
    const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63u;
    unsigned acc = 0;
    for (unsigned i = 0; i < iters; ++i) {
        unsigned b = 0;
        if (lane == 0) b = atomicAdd(ctr + ((wave + i) % k_words) * stride_words, 64u);
        b = __builtin_amdgcn_readfirstlane(b);
        acc += b;
    }
    if (acc == 0xFFFFFFFFu) sink[0] = acc;
}
int main()
{
    unsigned *ctr, *sink;
    hipMalloc(&ctr, 1 << 20); hipMalloc(&sink, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const unsigned grid = 256 * 7, iters = 2000;
    for (unsigned stride : { 1u, 16u, 64u, 1024u })
        for (unsigned kw : { 1u, 2u, 4u, 8u, 16u, 64u }) {
            hipMemset(ctr, 0, 1 << 20);
            hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, ctr, kw, stride, 10u, sink);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, ctr, kw, stride, iters, sink);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("words %3u stride %5u B: %8.1f M atomics/s\n", kw, stride * 4, (double)grid * 4 * iters / ms / 1e3);
        }
    return 0;
}
