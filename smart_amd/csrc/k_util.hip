// k_util.hip — corpus kernels: generate_text, tile_fill, text_alphabet, probe_read
// (one translation unit per kernel family: dev_common.hpp)
#include "dev_common.hpp"
#include "launch_common.hpp"

namespace sg {

// ---------------------------------------------------------------------------
// corpus kernels
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// dst[i] = corpus byte (off+i); one thread produces one aligned 8-byte corpus word
// (one 8-byte store when the word lies wholly inside the request and off%8==0).
__global__ __launch_bounds__(256) void generate_text(uint8_t* dst, uint64_t seed, uint32_t sigma,
                                                     uint64_t off, uint64_t n)
{
    const uint64_t w_first = off >> 3, w_last = (off + n + 7) >> 3;  // corpus words touched
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const bool pow2 = (sigma & (sigma - 1)) == 0;
    const bool aligned = (off & 7) == 0;
    for (uint64_t wi = w_first + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; wi < w_last;
         wi += stride) {
        const uint64_t x = splitmix64(seed + wi);
        uint64_t y;
        if (pow2) {
            y = x & (0x0101010101010101ull * (uint64_t)(sigma - 1));
        } else {
            y = 0;
#pragma unroll
            for (int b = 0; b < 8; ++b)
                y |= (uint64_t)(((uint32_t)(x >> (8 * b)) & 0xFFu) % sigma) << (8 * b);
        }
        const uint64_t j0 = wi << 3;  // corpus offset of byte 0 of this word
        if (aligned && j0 + 8 <= off + n) {
            *reinterpret_cast<uint64_t*>(dst + (j0 - off)) = y;
        } else {
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const uint64_t j = j0 + b;
                if (j >= off && j < off + n) dst[j - off] = (uint8_t)(y >> (8 * b));
            }
        }
    }
}

// dst[i] = unit[(phase + i) % unit_len]
__global__ __launch_bounds__(256) void tile_fill(uint8_t* dst, const uint8_t* unit,
                                                 uint64_t unit_len, uint64_t phase, uint64_t n)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        dst[i] = unit[(phase + i) % unit_len];
}

// Which byte values occur in the text (taken once, when a text is created — api.cpp text_alphabet): every workgroup
// marks them in LDS and ORs its 256 bits into out[8].
__global__ __launch_bounds__(256) void text_alphabet(const uint8_t* text, uint64_t n, uint32_t* out)
{
    __shared__ uint32_t seen[256];
    seen[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t n16 = n / 16, stride = (uint64_t)gridDim.x * 256;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) {
        const uint4 v = ld_stream16(text + 16 * i);
        const uint32_t d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 16; ++q) seen[(d[q >> 2] >> (8 * (q & 3))) & 0xFFu] = 1;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 15u)) seen[text[16 * n16 + threadIdx.x]] = 1;
    __syncthreads();
    const uint64_t bits = __ballot(seen[threadIdx.x] != 0);  // wave w of the workgroup: byte values 64w .. 64w+63
    if ((threadIdx.x & 63u) == 0) {
        if ((uint32_t)bits) atomicOr(out + 2 * (threadIdx.x >> 6), (uint32_t)bits);
        if ((uint32_t)(bits >> 32)) atomicOr(out + 2 * (threadIdx.x >> 6) + 1, (uint32_t)(bits >> 32));
    }
}

hipError_t launch_text_alphabet(const uint8_t* text, uint64_t n, uint32_t* out, int num_cus, hipStream_t stream)
{
    const uint64_t want = (n / 16 + 255) / 256;
    const uint32_t grid = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(want, (uint64_t)num_cus * 8));
    hipLaunchKernelGGL(text_alphabet, dim3(grid), dim3(256), 0, stream, text, n, out);
    return hipGetLastError();
}

// Streaming-read probe: the practical HBM read ceiling of this device for the
// access pattern the scan kernels use (coalesced 16 B/lane, 8 loads in flight per
// lane, every byte read once).  XOR-folds the text so the loads cannot be elided.
__global__ __launch_bounds__(256) void probe_read(const uint8_t* text, uint64_t n16,
                                                  unsigned long long* sink)
{
    const uint4* p = reinterpret_cast<const uint4*>(text);
    uint4 acc = {0, 0, 0, 0};
    const uint64_t stride = (uint64_t)gridDim.x * 256 * 8;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 * 8 + threadIdx.x; i < n16; i += stride) {
        uint4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k)
            v[k] = i + k * 256 < n16 ? ld_stream16(reinterpret_cast<const uint8_t*>(p + i + k * 256)) : uint4{0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 8; ++k) { acc.x ^= v[k].x; acc.y ^= v[k].y; acc.z ^= v[k].z; acc.w ^= v[k].w; }
    }
    const uint32_t f = acc.x ^ acc.y ^ acc.z ^ acc.w;
    if (f == 0x9E3779B9u) atomicAdd(sink, 1ull);  // practically never; keeps the loads live
}

hipError_t launch_probe_read(const uint8_t* text, uint64_t n, unsigned long long* sink, int num_cus,
                             hipStream_t stream)
{
    hipLaunchKernelGGL(probe_read, dim3((uint32_t)num_cus * 8), dim3(256), 0, stream, text, n / 16, sink);
    return hipGetLastError();
}

hipError_t launch_generate(uint8_t* dst, uint64_t seed, int sigma, uint64_t off, uint64_t n,
                           hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    const uint64_t words = ((off + n + 7) >> 3) - (off >> 3);
    uint64_t grid = (words + 255) / 256;
    if (grid > 65536) grid = 65536;
    hipLaunchKernelGGL(generate_text, dim3((uint32_t)grid), dim3(256), 0, stream, dst, seed,
                       (uint32_t)sigma, off, n);
    return hipGetLastError();
}

hipError_t launch_tile_fill(uint8_t* dst, const uint8_t* unit, uint64_t unit_len, uint64_t phase,
                            uint64_t n, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    uint64_t grid = (n + 255) / 256;
    if (grid > 65536) grid = 65536;
    hipLaunchKernelGGL(tile_fill, dim3((uint32_t)grid), dim3(256), 0, stream, dst, unit, unit_len,
                       phase, n);
    return hipGetLastError();
}


}  // namespace sg
