// VALU issue-rate probe (gfx950): cycles per wave64 instruction per SIMD for the integer ops the runs kernels
// are made of, at 1, 2 and 4 waves per SIMD.   hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int OP>
__global__ __launch_bounds__(1024) void k(uint32_t* out, int iters, unsigned long long* cyc)
{
    uint32_t a = threadIdx.x, b = threadIdx.x * 3 + 1, c = threadIdx.x ^ 0x55, d = 7 + threadIdx.x;
    uint32_t e = a + 1, f = b + 2, g = c + 3, h = d + 4;
    uint64_t q0 = a, q1 = b, q2 = c, q3 = d;
    const uint64_t src = ((uint64_t)h << 32) | g;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        // 8 independent chains x 8 = 64 instructions per REP64
        if (OP == 0) { REP8(asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(i));) }
        if (OP == 1) { REP8(asm volatile("v_perm_b32 %0, %0, %8, %1\n v_perm_b32 %1, %1, %8, %2\n v_perm_b32 %2, %2, %8, %3\n v_perm_b32 %3, %3, %8, %4\n v_perm_b32 %4, %4, %8, %5\n v_perm_b32 %5, %5, %8, %6\n v_perm_b32 %6, %6, %8, %7\n v_perm_b32 %7, %7, %8, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(i));) }
        if (OP == 2) { REP8(asm volatile("v_lshl_or_b32 %0, %0, 1, %8\n v_lshl_or_b32 %1, %1, 1, %8\n v_lshl_or_b32 %2, %2, 1, %8\n v_lshl_or_b32 %3, %3, 1, %8\n v_lshl_or_b32 %4, %4, 1, %8\n v_lshl_or_b32 %5, %5, 1, %8\n v_lshl_or_b32 %6, %6, 1, %8\n v_lshl_or_b32 %7, %7, 1, %8" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(i));) }
        if (OP == 3) { REP8(asm volatile("v_alignbit_b32 %0, %0, %8, 28\n v_alignbit_b32 %1, %1, %8, 28\n v_alignbit_b32 %2, %2, %8, 28\n v_alignbit_b32 %3, %3, %8, 28\n v_alignbit_b32 %4, %4, %8, 28\n v_alignbit_b32 %5, %5, %8, 28\n v_alignbit_b32 %6, %6, %8, 28\n v_alignbit_b32 %7, %7, %8, 28" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(i));) }
        if (OP == 4) { REP8(asm volatile("v_xor_b32 %0, %0, %8\n v_xor_b32 %1, %1, %8\n v_xor_b32 %2, %2, %8\n v_xor_b32 %3, %3, %8\n v_xor_b32 %4, %4, %8\n v_xor_b32 %5, %5, %8\n v_xor_b32 %6, %6, %8\n v_xor_b32 %7, %7, %8" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(i));) }
        if (OP == 5) { REP8(asm volatile("v_or3_b32 %0, %0, %8, %1\n v_or3_b32 %1, %1, %8, %2\n v_or3_b32 %2, %2, %8, %3\n v_or3_b32 %3, %3, %8, %4\n v_or3_b32 %4, %4, %8, %5\n v_or3_b32 %5, %5, %8, %6\n v_or3_b32 %6, %6, %8, %7\n v_or3_b32 %7, %7, %8, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(i));) }
        if (OP == 6) { REP8(asm volatile("v_lshlrev_b32 %0, 1, %0\n v_lshlrev_b32 %1, 1, %1\n v_lshlrev_b32 %2, 1, %2\n v_lshlrev_b32 %3, 1, %3\n v_lshlrev_b32 %4, 1, %4\n v_lshlrev_b32 %5, 1, %5\n v_lshlrev_b32 %6, 1, %6\n v_lshlrev_b32 %7, 1, %7" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(i));) }
        if (OP == 7) { REP8(asm volatile("v_pk_add_u16 %0, %0, %8\n v_pk_add_u16 %1, %1, %8\n v_pk_add_u16 %2, %2, %8\n v_pk_add_u16 %3, %3, %8\n v_pk_add_u16 %4, %4, %8\n v_pk_add_u16 %5, %5, %8\n v_pk_add_u16 %6, %6, %8\n v_pk_add_u16 %7, %7, %8" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(i));) }
        if (OP == 8) { REP8(asm volatile("v_max3_u32 %0, %0, %8, %1\n v_max3_u32 %1, %1, %8, %2\n v_max3_u32 %2, %2, %8, %3\n v_max3_u32 %3, %3, %8, %4\n v_max3_u32 %4, %4, %8, %5\n v_max3_u32 %5, %5, %8, %6\n v_max3_u32 %6, %6, %8, %7\n v_max3_u32 %7, %7, %8, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(i));) }
        if (OP >= 10 && OP <= 11) {  // 64-bit results: four independent chains through the accumulate operand
            REP8(REP8(asm volatile("v_mqsad_pk_u16_u8 %0, %4, %5, %0\n v_mqsad_pk_u16_u8 %1, %4, %5, %1\n v_mqsad_pk_u16_u8 %2, %4, %5, %2\n v_mqsad_pk_u16_u8 %3, %4, %5, %3\n"
                                   "v_mqsad_pk_u16_u8 %0, %4, %5, %0\n v_mqsad_pk_u16_u8 %1, %4, %5, %1\n v_mqsad_pk_u16_u8 %2, %4, %5, %2\n v_mqsad_pk_u16_u8 %3, %4, %5, %3"
                                   : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(src), "v"(i));))
        }
        if (OP == 12) { REP8(asm volatile("v_pk_min_u16 %0, %0, %8\n v_pk_min_u16 %1, %1, %8\n v_pk_min_u16 %2, %2, %8\n v_pk_min_u16 %3, %3, %8\n v_pk_min_u16 %4, %4, %8\n v_pk_min_u16 %5, %5, %8\n v_pk_min_u16 %6, %6, %8\n v_pk_min_u16 %7, %7, %8" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(i));) }
        if (OP == 9) { REP8(asm volatile("v_bfe_u32 %0, %0, 8, 8\n v_bfe_u32 %1, %1, 8, 8\n v_bfe_u32 %2, %2, 8, 8\n v_bfe_u32 %3, %3, 8, 8\n v_bfe_u32 %4, %4, 8, 8\n v_bfe_u32 %5, %5, 8, 8\n v_bfe_u32 %6, %6, 8, 8\n v_bfe_u32 %7, %7, 8, 8" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(i));) }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ e ^ f ^ g ^ h ^ (uint32_t)(q0 ^ q1 ^ q2 ^ q3) ^ (uint32_t)((q0 ^ q1 ^ q2 ^ q3) >> 32);
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int OP>
void run(const char* name, uint32_t* out, unsigned long long* cyc)
{
    const int iters = 2000;  // 64 instructions per iteration
    printf("%-16s", name);
    for (int threads : {256, 512, 1024}) {  // 1, 2, 4 waves per SIMD on one workgroup per CU
        hipLaunchKernelGGL(k<OP>, dim3(256), dim3(threads), 0, 0, out, iters, cyc);
        hipDeviceSynchronize();
        unsigned long long c = 0;
        hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        const double per_wave = (double)c / (iters * 64.0);
        printf("  %d waves/SIMD: %5.2f cyc/instr/wave = %5.2f cyc/instr/SIMD", threads / 256, per_wave, per_wave / (threads / 256));
    }
    printf("\n");
}

int main()
{
    uint32_t* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 8);
    run<0>("v_add_u32", out, cyc); run<4>("v_xor_b32", out, cyc); run<6>("v_lshlrev_b32", out, cyc);
    run<1>("v_perm_b32", out, cyc); run<2>("v_lshl_or_b32", out, cyc); run<3>("v_alignbit_b32", out, cyc);
    run<5>("v_or3_b32", out, cyc); run<8>("v_max3_u32", out, cyc); run<9>("v_bfe_u32", out, cyc); run<7>("v_pk_add_u16", out, cyc);
    run<12>("v_pk_min_u16", out, cyc); run<10>("v_mqsad_pk_u16", out, cyc);
    return 0;
}
