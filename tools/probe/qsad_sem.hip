// v_mqsad_pk_u16_u8 / v_qsad_pk_u16_u8 on gfx950 against a host model (which byte is which, what the mask means, how the
// accumulator joins):   hipcc --offload-arch=gfx950 -O3 -o qsad_sem qsad_sem.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>

__global__ void k(const uint64_t* s0, const uint32_t* s1, const uint64_t* s2, uint64_t* q, uint64_t* mq, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    q[i] = __builtin_amdgcn_qsad_pk_u16_u8(s0[i], s1[i], s2[i]);
    mq[i] = __builtin_amdgcn_mqsad_pk_u16_u8(s0[i], s1[i], s2[i]);
}

static uint64_t model(uint64_t s0, uint32_t s1, uint64_t s2, bool masked)
{
    uint64_t out = 0;
    for (int f = 0; f < 4; ++f) {
        const uint32_t win = (uint32_t)(s0 >> (8 * f));
        uint32_t sad = (uint32_t)((s2 >> (16 * f)) & 0xFFFF);
        for (int b = 0; b < 4; ++b) {
            const int x = (win >> (8 * b)) & 0xFF, r = (s1 >> (8 * b)) & 0xFF;
            if (masked && r == 0) continue;
            sad += abs(x - r);
        }
        if (sad > 0xFFFF) sad = 0xFFFF;
        out |= (uint64_t)sad << (16 * f);
    }
    return out;
}

int main()
{
    const int n = 1 << 16;
    std::vector<uint64_t> s0(n), s2(n), q(n), mq(n);
    std::vector<uint32_t> s1(n);
    srand(7);
    auto r64 = [] { uint64_t x = 0; for (int i = 0; i < 8; ++i) x = (x << 8) | (rand() & 0xFF); return x; };
    for (int i = 0; i < n; ++i) {
        s0[i] = r64();
        s1[i] = (uint32_t)r64();
        if (i % 3 == 0) s1[i] &= 0x0000FFFFu;              // masked-out reference bytes
        if (i % 5 == 0) s1[i] &= 0xFF00FFFFu;
        s2[i] = i % 2 ? 0 : (r64() & 0x0FFF0FFF0FFF0FFFull);
        if (i % 7 == 0) { s1[i] = (uint32_t)(s0[i] >> (8 * (i % 4))); }  // an exact match at one alignment
    }
    uint64_t *d0, *d2, *dq, *dm; uint32_t* d1;
    hipMalloc(&d0, n * 8); hipMalloc(&d2, n * 8); hipMalloc(&dq, n * 8); hipMalloc(&dm, n * 8); hipMalloc(&d1, n * 4);
    hipMemcpy(d0, s0.data(), n * 8, hipMemcpyHostToDevice);
    hipMemcpy(d1, s1.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(d2, s2.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, d0, d1, d2, dq, dm, n);
    hipMemcpy(q.data(), dq, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(mq.data(), dm, n * 8, hipMemcpyDeviceToHost);
    int badq = 0, badm = 0;
    for (int i = 0; i < n; ++i) {
        if (q[i] != model(s0[i], s1[i], s2[i], false) && badq++ < 4)
            printf("qsad  s0 %016llx s1 %08x s2 %016llx -> %016llx, model %016llx\n", (unsigned long long)s0[i], s1[i], (unsigned long long)s2[i], (unsigned long long)q[i], (unsigned long long)model(s0[i], s1[i], s2[i], false));
        if (mq[i] != model(s0[i], s1[i], s2[i], true) && badm++ < 4)
            printf("mqsad s0 %016llx s1 %08x s2 %016llx -> %016llx, model %016llx\n", (unsigned long long)s0[i], s1[i], (unsigned long long)s2[i], (unsigned long long)mq[i], (unsigned long long)model(s0[i], s1[i], s2[i], true));
    }
    printf("qsad mismatches %d, mqsad mismatches %d of %d\n", badq, badm, n);
    return 0;
}
