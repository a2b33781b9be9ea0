// LDS gather-rate probe (gfx950): cycles per wave64 LDS instruction per CU with per-lane random addresses, as the
// runs kernels issue them (bank-private b32 gathers, u8 automaton lookups), 16 waves per CU, 8 independent gathers
// in flight per wave.   hipcc --offload-arch=gfx950 -O3 -o lds_gather lds_gather.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef __attribute__((address_space(3))) uint32_t lds_u32_t;
typedef __attribute__((address_space(3))) uint8_t lds_u8_t;
typedef __attribute__((address_space(3))) uint16_t lds_u16_t;
typedef __attribute__((address_space(3))) uint64_t lds_u64_t;

// MODE 0: ds_read_b32, bank-private (addr = c*256 + lane*4)   1: ds_read_b32, shared 1 KB table (addr = c*4)
//      2: ds_read_u8 from a 64 KB table, random                3: ds_read_u16 bank-private   4: ds_read_b64 bank-private (c*512+lane*8 -> 128 KB)
//      5: ds_read_b32 same address in all lanes (broadcast)    6: ds_read_b128 contiguous (lane*16)
template <int MODE>
__global__ __launch_bounds__(1024) void k(uint32_t* out, int iters, unsigned long long* cyc, uint32_t seed)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    for (uint32_t i = threadIdx.x; i < 160 * 1024 / 4; i += blockDim.x) reinterpret_cast<uint32_t*>(smem)[i] = i * 2654435761u;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t x = seed + threadIdx.x * 747796405u, acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        uint32_t r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            x = x * 1664525u + 1013904223u;
            const uint32_t c = (x >> 13) & 0xFFu;
            uint32_t addr;
            if (MODE == 0) addr = c * 256u + lane * 4u;
            else if (MODE == 1) addr = c * 4u;
            else if (MODE == 2) addr = (x >> 9) & 0xFFFFu;
            else if (MODE == 3) addr = c * 256u + lane * 4u;
            else if (MODE == 4) addr = c * 512u + lane * 8u;
            else if (MODE == 5) addr = (uint32_t)(i & 255) * 4u;
            else addr = lane * 16u + (uint32_t)(j * 1024);
            if (MODE == 2) r[j] = *(const lds_u8_t*)(size_t)addr;
            else if (MODE == 3) r[j] = *(const lds_u16_t*)(size_t)addr;
            else if (MODE == 4) { const uint64_t v = *(const lds_u64_t*)(size_t)addr; r[j] = (uint32_t)v ^ (uint32_t)(v >> 32); }
            else if (MODE == 6) { const uint4 v = *reinterpret_cast<const uint4*>(smem + addr); r[j] = v.x ^ v.y ^ v.z ^ v.w; }
            else r[j] = *(const lds_u32_t*)(size_t)addr;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc ^= r[j];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int MODE>
void run(const char* name, uint32_t* out, unsigned long long* cyc)
{
    const int iters = 4000;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    printf("%-44s", name);
    for (int threads : {256, 512, 1024}) {
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 160 * 1024, 0, out, iters, cyc, 12345u);
        hipDeviceSynchronize();
        unsigned long long c = 0;
        hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        const double per_cu = (double)c / (iters * 8.0 * (threads / 64));
        printf("  %2d waves/CU: %5.2f clk per wave-instr per CU (%5.1f lanes/clk)", threads / 64, per_cu, 64.0 / per_cu);
    }
    printf("\n");
}

int main()
{
    uint32_t* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 8);
    run<0>("ds_read_b32 bank-private gather (so_runs)", out, cyc);
    run<1>("ds_read_b32 shared 1 KB table gather", out, cyc);
    run<2>("ds_read_u8 random in 64 KB (kmp_runs)", out, cyc);
    run<3>("ds_read_u16 bank-private gather", out, cyc);
    run<4>("ds_read_b64 bank-private gather", out, cyc);
    run<5>("ds_read_b32 broadcast", out, cyc);
    run<6>("ds_read_b128 contiguous", out, cyc);
    return 0;
}
