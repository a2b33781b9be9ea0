// Streaming-read ceiling probe: variants of a read-and-fold kernel over 1 GiB.
//   hipcc --offload-arch=gfx950 -O3 -o read_bw read_bw.hip && ./read_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void rd(const uint4* __restrict__ p, size_t n16, unsigned long long* sink)
{
    uint4 acc = {0, 0, 0, 0};
    const size_t stride = (size_t)gridDim.x * 256 * UNROLL;
    for (size_t i = (size_t)blockIdx.x * 256 * UNROLL + threadIdx.x; i < n16; i += stride) {
        uint4 v[UNROLL];
#pragma unroll
        for (int k = 0; k < UNROLL; ++k) {
            const uint4* q = p + (i + (size_t)k * 256 < n16 ? i + (size_t)k * 256 : i);
            if (NT) {
                v[k].x = __builtin_nontemporal_load(&q->x); v[k].y = __builtin_nontemporal_load(&q->y);
                v[k].z = __builtin_nontemporal_load(&q->z); v[k].w = __builtin_nontemporal_load(&q->w);
            } else v[k] = *q;
        }
#pragma unroll
        for (int k = 0; k < UNROLL; ++k) { acc.x ^= v[k].x; acc.y ^= v[k].y; acc.z ^= v[k].z; acc.w ^= v[k].w; }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9E3779B9u) atomicAdd(sink, 1ull);
}

template <int UNROLL, bool NT>
double run(const uint4* p, size_t n16, unsigned long long* sink, int grid)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((rd<UNROLL, NT>), dim3(grid), dim3(256), 0, 0, p, n16, sink);
    hipEventRecord(a);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((rd<UNROLL, NT>), dim3(grid), dim3(256), 0, 0, p, n16, sink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return (double)n16 * 16 * 20 / (ms * 1e-3) / 1e9;
}

int main()
{
    const size_t n = 1ull << 30;
    uint4* p; unsigned long long* sink;
    CHECK(hipMalloc(&p, n)); CHECK(hipMalloc(&sink, 8));
    CHECK(hipMemset(p, 1, n)); CHECK(hipMemset(sink, 0, 8));
    const size_t n16 = n / 16;
    for (int grid : {1024, 2048, 4096, 8192, 16384}) {
        printf("grid %5d  u4 %7.1f  u8 %7.1f  u16 %7.1f | nt: u4 %7.1f  u8 %7.1f  u16 %7.1f GB/s\n", grid,
               run<4, false>(p, n16, sink, grid), run<8, false>(p, n16, sink, grid), run<16, false>(p, n16, sink, grid),
               run<4, true>(p, n16, sink, grid), run<8, true>(p, n16, sink, grid), run<16, true>(p, n16, sink, grid));
    }
    return 0;
}
