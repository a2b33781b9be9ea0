// k_ab.hip — the superseded scan kernels of kernels_ab.inc and their launchers: ONLY in the A/B build
// (make AB=1 -> libsmartgpu_ab.so, -DSMARTGPU_AB), selected by smartgpu_tune.  The product library does not carry them.
// (one translation unit per kernel family: dev_common.hpp)
#ifdef SMARTGPU_AB
#include "dev_common.hpp"
#include "runs_common.hpp"
#include "launch_common.hpp"

namespace sg {

constexpr int kSoT = 256, kSoL = 80;  // so_scan

#include "kernels_ab.inc"

// tune(6, 1 / 2 / 4), Shift-And's own AND form (tune(6,3)): the earlier Shift-Or kernels.  a.so_off: prepare_scan_args.
hipError_t launch_ab_so(int algo, const ScanArgs& a, int num_cus, hipStream_t stream, bool* handled)
{
    const uint32_t m = a.m;
    *handled = true;
    if (g_tune[6] == 2 && algo == SMARTGPU_SO) {  // the first runs kernel: shared table, 64-byte steps
        uint64_t L = g_tune[5] ? (uint64_t)g_tune[5] : 1024;
        const uint64_t fill = (a.s_end - a.s_begin) / ((uint64_t)num_cus * 16 * 64);
        if (L > fill) L = fill;
        if (L < 256) L = 256;
        L = (L + 63) & ~63ull;
        const TileRange tr = tiles_for(a.s_begin, a.s_end, L);
        if (tr.count == 0) return hipSuccess;
        uint64_t grid = ((uint64_t)tr.count + 255) / 256;
        const uint64_t cap = (uint64_t)num_cus * 6;
        if (grid > cap) grid = cap;
        const size_t lds = 1040 + 4 * (size_t)kRunSlab;
        if (m > 32)
            hipLaunchKernelGGL(so_runs64<true>, dim3((uint32_t)grid, g_batch.count), dim3(256), lds, stream, a, (uint32_t)L, (uint64_t)tr.count, g_batch.items);
        else
            hipLaunchKernelGGL(so_runs64<false>, dim3((uint32_t)grid, g_batch.count), dim3(256), lds, stream, a, (uint32_t)L, (uint64_t)tr.count, g_batch.items);
        return hipGetLastError();
    }
    if (g_tune[6] == 1 && algo == SMARTGPU_SO) {  // LDS tiles
        const size_t lds = 1040 + (size_t)kSoT * kSoL + 32;
        const TileRange tr = tiles_for(a.s_begin, a.s_end, (uint64_t)kSoT * kSoL);
        if (m > 32) return launch_tiled(so_scan<kSoT, kSoL, true>, a, tr, kSoT, lds, 6, num_cus, stream);
        return launch_tiled(so_scan<kSoT, kSoL, false>, a, tr, kSoT, lds, 6, num_cus, stream);
    }
    const bool shift_and = algo == SMARTGPU_SA && g_tune[6] == 3;
    if (shift_and || g_tune[6] == 4) {  // so_runs1: a step per byte behind the whole-line loader (LineIo); launch_so_runs' geometry
        const uint64_t lmin = g_tune[5] ? std::min<uint64_t>((uint64_t)g_tune[5], kRunLenMax / 2) : 2048;
        const uint64_t L = balanced_run_len(a.s_begin, a.s_end, 64, (uint64_t)num_cus * kRunWaves, lmin, 2 * lmin, 128);
        const TileRange tr = tiles_for(a.s_begin, a.s_end, L);
        if (tr.count == 0) return hipSuccess;
        const size_t lds = 65536 + kRunWaves * (size_t)kLineSlab;
        const uint64_t grid = runs_grid(tr.count, num_cus);
        trace_runs("so_runs1", a, L, tr, grid);
#define SG_SO_RUNS1(L_, A_)                                                                               \
    do {                                                                                                 \
        allow_lds(reinterpret_cast<const void*>(so_runs1<L_, A_>), lds);                                  \
        hipLaunchKernelGGL((so_runs1<L_, A_>), dim3((uint32_t)grid, g_batch.count), dim3(64 * kRunWaves), lds, stream, a, \
                           (uint32_t)L, (uint64_t)tr.count, g_batch.items);                              \
    } while (0)
        if (shift_and) { if (m > 32) SG_SO_RUNS1(true, true); else SG_SO_RUNS1(false, true); }
        else { if (m > 32) SG_SO_RUNS1(true, false); else SG_SO_RUNS1(false, false); }
#undef SG_SO_RUNS1
        return hipGetLastError();
    }
    *handled = false;
    return hipSuccess;
}

// tune(3, 1 / 2 / 3): kmp_scan (LDS tiles, failure links), kmp_links_runs, kmp_runs1
hipError_t launch_ab_kmp(const ScanArgs& a, int num_cus, hipStream_t stream, bool* handled)
{
    const uint32_t m = a.m;
    *handled = true;
    if (g_tune[3] == 1 && m <= 40) {
        // LDS tiles, run length per lane 80 or 144 bytes (re-scan of m-1 bytes <~28 %), failure links
        const size_t fixed = r16(4 * m) + r16(m - 1);
#define SG_KMP(T_, L_, WGS_)                                                                   \
    do {                                                                                       \
        const size_t lds = fixed + (size_t)(T_) * (L_);                                        \
        if (lds > 64 * 1024) allow_lds(reinterpret_cast<const void*>(kmp_scan<T_, L_>), lds);  \
        const TileRange tr = tiles_for(a.s_begin, a.s_end, (uint64_t)(T_) * (L_));            \
        return launch_tiled(kmp_scan<T_, L_>, a, tr, T_, lds, WGS_, num_cus, stream);          \
    } while (0)
        if (m <= 16) SG_KMP(256, 80, 6);
        SG_KMP(256, 144, 4);
#undef SG_KMP
    }
    const bool links = g_tune[3] == 2;  // failure links
    const bool v1 = g_tune[3] == 3;     // the previous kernel (running maximum, half-line loader): its table follows kmp_runs'
    if (!links && !v1) { *handled = false; return hipSuccess; }
    const uint32_t dfa_off = kmp_spread_off(m);  // the spread table of kmp_runs (api.cpp build_blob); kmp_runs1's follows it
    const uint32_t w0 = a.prefer_packed ? a.prefer_packed : kmp_window(m);
    const uint32_t dfa1_off = dfa_off + (w0 < 63 ? w0 + 1 : 256u) * 256 + kKmpQBytes;
    const uint32_t w = (links || m <= kKmpDfaMaxM) ? m : kKmpDfaMaxM;
    if (links) {
        const uint64_t span = a.s_end - a.s_begin;
        uint64_t L = 8ull * (w - 1);
        const uint64_t fill = span / ((uint64_t)num_cus * 16 * 64);
        if (L > fill) L = fill;
        if (L < 2ull * (w - 1)) L = 2ull * (w - 1);
        if (L < 512) L = 512;
        L = (L + 63) & ~63ull;
        const TileRange tr = tiles_for(a.s_begin, a.s_end, L);
        if (tr.count == 0) return hipSuccess;
        const size_t lds = r16(4 * m) + 4 * (size_t)kRunSlab;
        uint64_t grid = ((uint64_t)tr.count + 255) / 256;
        const uint64_t cap = (uint64_t)num_cus * 4;
        if (grid > cap) grid = cap;
        hipLaunchKernelGGL(kmp_links_runs, dim3((uint32_t)grid, g_batch.count), dim3(256), lds, stream, a, (uint32_t)L,
                           (uint64_t)tr.count, dfa1_off, g_batch.items);
        return hipGetLastError();
    }
    const size_t table = (size_t)(w < 64 ? 4 * w + 1 : 256) * 256;  // kmp_runs1: rows up to the accept id
    const size_t lds = table + kKmpQBytes + kRunWaves * (size_t)kLineSlab;
    uint64_t lmin = g_tune[5] ? std::min<uint64_t>((uint64_t)g_tune[5], kRunLenMax / 2) : 2048;
    if (lmin < 8ull * (w - 1)) lmin = 8ull * (w - 1);
    const uint64_t lfloor = 2ull * (w - 1) > 128 ? 2ull * (w - 1) : 128;
    const uint64_t L = balanced_run_len(a.s_begin, a.s_end, 64, (uint64_t)num_cus * kRunWaves, lmin, 2 * lmin, lfloor);
    const TileRange tr = tiles_for(a.s_begin, a.s_end, L);
    if (tr.count == 0) return hipSuccess;
    const uint64_t grid = runs_grid(tr.count, num_cus, kRunWaves);
    trace_runs("kmp_runs1", a, L, tr, grid);
#define SG_KMP_RUNS(K_, OFF_)                                                                            \
    do {                                                                                                 \
        if (lds > 64 * 1024) allow_lds(reinterpret_cast<const void*>(K_), lds);                          \
        hipLaunchKernelGGL(K_, dim3((uint32_t)grid, g_batch.count), dim3(64 * kRunWaves), lds, stream, a, (uint32_t)L, \
                           (uint64_t)tr.count, (uint32_t)(OFF_), g_batch.items);                         \
    } while (0)
    if (m > kKmpDfaMaxM) SG_KMP_RUNS(kmp_runs1<true>, dfa1_off); else SG_KMP_RUNS(kmp_runs1<false>, dfa1_off);
#undef SG_KMP_RUNS
    return hipGetLastError();
}

}  // namespace sg
#endif  // SMARTGPU_AB
