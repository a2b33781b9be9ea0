// launch_common.hpp — host-side launch helpers shared by the kernel translation units (k_*.hip) and the dispatcher
// (launch.hip): tile / run geometry, the grid shapes, the per-family launch entry points.
#pragma once
#include "kernels.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <utility>

namespace sg {

struct TileRange { uint64_t first; uint32_t count; };
// A pattern set in ONE grid: while set, every scan launch uses gridDim.y = count and hands the kernels the
// device array of per-pattern arguments (they take argument set blockIdx.y instead of the by-value one).
struct BatchCtx { const BatchItem* items; uint32_t count; };
extern thread_local BatchCtx g_batch;  // launch.hip

// tiles of `tb` absolute offsets intersecting [lo, hi)
inline TileRange tiles_for(uint64_t lo, uint64_t hi, uint64_t tb)
{
    if (hi <= lo) return {0, 0};
    const uint64_t first = lo / tb, last = (hi - 1) / tb;
    return {first, (uint32_t)(last - first + 1)};
}

inline uint32_t r16(uint32_t x) { return (x + 15u) & ~15u; }

// Workgroups per CU of the LDS-tile skip kernels.  FOUR (64 KB of tiles in flight per CU), not
// the eight or nine the LDS would hold, when the pattern promises a pure streaming scan
// (a.sparse, api.cpp): measured on 1, 1.37 and 4 GiB of rand128, HOR m=32 runs at 85-87 % of
// 8 TB/s with 4, 78-82 % with 8, 75 % with 6, 82 % with 16 (two rounds) — profiles/r01/
// o_wgs_per_cu.log; BM and BNDM follow the same curve.  Where lanes spend their time verifying
// (English text: HOR m=64 47 % with 8, 40 % with 4) the extra waves pay: EIGHT.
// Short windows (8 <= m < 16) of such patterns: FIVE (HOR m = 8..13: 75-83 % with 5, 74-83 % with 6, 68-83 %
// with 4, 71-74 % with 8).  bm_scan, with its larger tables: THREE for m >= 16 (83-84 % against 79-82 % with 4).
inline int tile_wgs(const ScanArgs& a, bool bm = false)
{
    if (!a.sparse) return 8;
    if (a.m < 16) return bm ? 4 : 5;  // bm_scan m = 12, 14: 78-80 % with 4, 75 % with 5
    return bm ? 3 : 4;
}

template <typename K>
hipError_t launch_tiled(K kernel, const ScanArgs& a, TileRange tr, int threads, size_t lds,
                        int wgs_per_cu, int num_cus, hipStream_t stream)
{
    if (tr.count == 0) return hipSuccess;
    if (g_tune[4]) wgs_per_cu = g_tune[4];  // A/B: workgroups per CU of the tile kernels
    uint32_t grid = (uint32_t)num_cus * (uint32_t)wgs_per_cu;
    if (grid > tr.count) grid = tr.count;
    hipLaunchKernelGGL(kernel, dim3(grid, g_batch.count), dim3(threads), lds, stream, a, tr.first, tr.count, g_batch.items);
    return hipGetLastError();
}

// tile shapes (threads, bytes per lane)
constexpr int kHorT = 256, kHorL = 64;
constexpr int kBmT = 256, kBmL = 64;
constexpr int kBmBusyT = 128;  // bm_scan / hor_flat where windows survive (English, small alphabets): two-wave workgroups, 12 per CU
constexpr int kBndmT = 256, kBndmL = 64;
constexpr int kBndmBusyT = 128;  // bndm_scan where windows survive: two-wave workgroups
constexpr int kEpsmT = 256;

// ---- the runs kernels (so_runs, kmp_runs) -----------------------------------------------------------------------
constexpr uint32_t kRunLine = 128;       // bytes of a run fetched per step
// The swap loader reads whole lines of 8 runs per instruction and up to 7 runs + a few lines past the text's last run
// (blocks past the last run re-read block 0, a partial block does not): that over-read must stay inside the text's
// back pad, which bounds the run length — smartgpu_tune(5, .) is clamped to it (launch_so_runs, launch_kmp_runs).
constexpr uint64_t kRunLenMax = 16384;   // runs of at most 16 KiB (the default: 2-4 KiB, 8 x 254 for the longest KMP window)
static_assert(8 * kRunLenMax + 512 + 4200 <= kBackPad, "the loaders' over-read past the last run must stay inside the back pad");
constexpr int kLineSlab = 64 * 64;       // LDS bytes per wave
constexpr int kRunWaves = 16;            // one 1024-thread workgroup per CU shares the table
// kmp_runs<., FOUR>: TWELVE.  Its four-byte rows are looked up with a random 8-bit index — bank conflicts in 40 % of its LDS
// cycles (profiles/r03/j_pmc_four_rand4_m32.txt) — and three waves per SIMD queue less behind one another on them than four:
// sigma = 2 / 4, m = 16 ... 1024, build against build: 12 waves -5 ... -9 % time, 10 the same, 14 +2 %, 8 +28 %; the byte-wise
// kernels lose with fewer (rand128: so_runs +5 %, kmp_runs +11 % at 12 waves), so_runs<., FOUR> is indifferent (-1 ... +2 %).
constexpr int kKmpFourWaves = 12;
// kmp_runs<., false, COMPACT> (k_kmp.hip): FIVE four-wave workgroups per CU — 20 waves where the 1024-thread form has 16 —
// each with its own compact table of the pattern or its 56-byte prefix: 5 x ((56 + 2) x 256 + 272 + 4 x 4096) = 157,520 of the
// CU's 163,840 bytes of LDS.
constexpr int kKmpCompactWaves = 4, kKmpCompactPerCu = 5;
static_assert(kKmpCompactPerCu * ((kKmpCompactWindow + 2) * 256 + kKmpQBytes + kKmpCompactWaves * kLineSlab) <= 160 * 1024, "five compact workgroups per CU");

// Run length for the runs kernels: every wave should get the same number of groups (`per_group`
// runs each), or the slowest wave sets the kernel time (12 waves/CU on 4096 groups: 67 %).
// Picks the smallest k such that span / (k * nwaves groups) gives runs of at most `lmax` bytes,
// then grows L in 128-byte steps until the runs cut on absolute offsets fit k * nwaves groups.
// Runs shorter than `lmin` (small texts) are not worth balancing: L = lmin.
inline uint64_t balanced_run_len(uint64_t s_begin, uint64_t s_end, uint64_t per_group, uint64_t nwaves,
                                 uint64_t lmin, uint64_t lmax, uint64_t lfloor)
{
    const uint64_t span = s_end - s_begin;
    const uint64_t slots = per_group * nwaves;  // runs per round of all waves
    const uint64_t k = (span + slots * lmax - 1) / (slots * lmax);
    // Runs are cut on absolute offsets and fetched 128 bytes at a time: a length that is not a multiple of 128
    // puts every second run's fetches across two memory lines (measured: 0.97 GiB, runs of 4032 bytes, 57 % where
    // 0.9 and 1.0 GiB — 3712 and 4096 — reach 79-82 %).
    uint64_t L = ((span + slots * k - 1) / (slots * k) + 127) & ~127ull;
    // A text too small to give every wave a group of lmin-byte runs (SMART's stock 1 MiB texts: 8 groups of
    // 2 KiB runs = 8 waves on the whole chip, 60-70 us per search): shorter runs, down to lfloor — the
    // re-scan of w-1 bytes per run costs less than the idle CUs.
    if (L < lmin) return L > lfloor ? L : (lfloor + 127) & ~127ull;
    while (tiles_for(s_begin, s_end, L).count > slots * k) L += 128;
    return L;
}

// Workgroups per pattern of the runs kernels (one 1024-thread workgroup per CU, first_group hands every workgroup a
// contiguous share of the groups of 64 runs).  One pattern: as many workgroups as there are groups, up to `per_cu` per CU —
// a small text is spread over the chip, one wave per CU.  A pattern set in one grid (gridDim.y patterns, SMART's -pset
// loop on its 1 MiB texts): every workgroup copies its pattern's table (up to 64 KB) before it starts, so with enough
// patterns to fill the chip the groups are packed 16 to a workgroup — 500 patterns x 128 groups of 128-byte runs:
// 4000 workgroups with all waves busy instead of 64000 with one (KMP 4.3 -> 0.6 us per pattern, measured).
inline uint64_t runs_grid(uint64_t nruns, int num_cus, int waves = kRunWaves, int per_cu = 1)
{
    const uint64_t groups = (nruns + 63) / 64;
    const uint64_t cap = (uint64_t)num_cus * per_cu;
    const uint64_t spread = groups < cap ? groups : cap;
    const uint64_t packed = (groups + waves - 1) / waves;
    uint64_t want = (2ull * cap + g_batch.count - 1) / g_batch.count;  // enough workgroups for two rounds of the chip
    if (want < packed) want = packed;
    return want < spread ? want : spread;
}

// SMARTGPU_DEBUG=1 in the environment: the geometry of every launch of a runs kernel on stderr (tools/nvar_probe.py)
inline void trace_runs(const char* kernel, const ScanArgs& a, uint64_t run_len, const TileRange& tr, uint64_t grid)
{
    static const bool on = getenv("SMARTGPU_DEBUG") != nullptr;
    if (on)
        fprintf(stderr, "%s: starts [%llu, %llu) runs of %llu bytes: %llu from run %llu, %llu workgroups x %u patterns\n", kernel,
                (unsigned long long)a.s_begin, (unsigned long long)a.s_end, (unsigned long long)run_len,
                (unsigned long long)tr.count, (unsigned long long)tr.first, (unsigned long long)grid, g_batch.count);
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (device, kernel), not per launch
inline void allow_lds(const void* kernel, size_t lds)
{
    static std::map<std::pair<int, const void*>, size_t> done;
    int dev = 0;
    (void)hipGetDevice(&dev);
    size_t& have = done[{dev, kernel}];
    if (have >= lds) return;
    (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    have = lds;
}

// ---- the families' launch entry points (each defined next to its kernels) ---------------------------------------
// k_hor.hip — a.halo = H; the packed regime is decided by the dispatcher (launch.hip) before these are called
hipError_t launch_hor(const ScanArgs& a, uint32_t q, int num_cus, hipStream_t stream);     // HOR, TUNEDBM: hor_scan<.., 0> / the flat form <.., 9>; q: <.., q>, the q-gram table
hipError_t launch_hor_var(int algo, const ScanArgs& a, int num_cus, hipStream_t stream);   // RAITA, QS, HASH3/5/8
hipError_t launch_kr(const ScanArgs& a, int num_cus, hipStream_t stream);                  // Karp-Rabin on the bank-private tiles
#ifdef SMARTGPU_AB
hipError_t launch_hor_bp(const ScanArgs& a, int num_cus, hipStream_t stream);              // Horspool on the bank-private tiles (tune(0,2))
#endif
// k_horg.hip — Horspool on grams (a text of at most four byte values): gram = 1: eight one-bit symbols, 2: four two-bit symbols
hipError_t launch_hor_gram(const ScanArgs& a, int gram, int num_cus, hipStream_t stream);
hipError_t launch_bm_gram(const ScanArgs& a, int gram, int num_cus, hipStream_t stream);
// k_bm.hip
hipError_t launch_bm(const ScanArgs& a, int num_cus, hipStream_t stream);
// k_bndm.hip, k_bndmx.hip
hipError_t launch_bndm(const ScanArgs& a, int num_cus, hipStream_t stream, TextCodes codes);
hipError_t launch_sbndm(const ScanArgs& a, int num_cus, hipStream_t stream);
hipError_t launch_bndml(const ScanArgs& a, int num_cus, hipStream_t stream);
// k_so.hip, k_kmp.hip
hipError_t launch_so_runs(const ScanArgs& a, bool shift_and, int num_cus, hipStream_t stream, TextCodes codes);
hipError_t launch_kmp_runs(const ScanArgs& a, int num_cus, hipStream_t stream, TextCodes codes);
// k_packed.hip — kind: which algorithm's plan the fingerprint belongs to (SMARTGPU_HOR / _BM / _BNDM / _EPSM: its verification tail)
hipError_t launch_packed(int kind, const ScanArgs& a, int num_cus, hipStream_t stream, TextCodes codes = TextCodes());
#ifdef SMARTGPU_AB
// k_ab.hip — the superseded kernels; *handled = false: the tune settings ask for none of them
hipError_t launch_ab_so(int algo, const ScanArgs& a, int num_cus, hipStream_t stream, bool* handled);
hipError_t launch_ab_kmp(const ScanArgs& a, int num_cus, hipStream_t stream, bool* handled);
#endif

}  // namespace sg
