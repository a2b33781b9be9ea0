// launch.hip — the dispatcher between the C-ABI layer (api.cpp) and the kernel families (k_*.hip): which kernel a
// plan's fields lead to (launch_scan, scan_kernel_name), the tuning knobs, pattern sets in one grid.  No kernel here.
//
// Every kernel computes the same function
//     count(P,T) = |{ s in [s_begin, s_end) : T[s..s+m) == P }|
// (overlapping occurrences count) and differs in the per-lane scan strategy and in the tables it stages in LDS.
// Three families, each family's kernels in translation units of their own (dev_common.hpp says why):
//
//  1. LDS TILES — the skip algorithms HOR, BM, BNDM (k_hor.hip, k_bm.hip, k_bndm.hip, k_bndmx.hip).
//     The text is cut into tiles of TB = THREADS*L bytes on ABSOLUTE text offsets, so
//     every tile load is 16-byte aligned and coalesced (global_load_dwordx4 nt, 1 KiB
//     per wave-load); tile t+1 is prefetched into registers while the lanes walk tile
//     t.  Tiles are indexed by window END position e = s+m-1 with a BACK halo of
//     H = min(m-1, 16) bytes: the byte that drives the shift, T[e], is always in the
//     tile; a lane verifies right-to-left through the halo by itself and — only for
//     m-1 > H and only after H+1 bytes matched — parks the window for a
//     wave-cooperative comparison of the rest (wave_verify).
//  2. RUNS THROUGH LDS SLABS — the serial automata SO and KMP (k_so.hip, k_kmp.hip).
//     A lane owns a run of 2-4 KiB of start positions (128+ bytes on small texts); the wave
//     fetches the next 128-byte line of each of its 64 runs with coalesced non-temporal loads,
//     parks it in its own LDS slab one 64-byte half at a time and every lane reads its run
//     back.  No workgroup barrier after the table set-up.
//  3. PACKED — EPSM, and the short-pattern / tiny-shift regime of the skip algorithms
//     (k_packed.hip): every alignment is compared from registers, no LDS.
//
// Restarting an algorithm at a lane / tile / run / GPU boundary preserves the count
// (SURVEY.md §7 restart table): skip algorithms carry no state between windows, the
// automata restart in their initial state and re-scan w-1 bytes.  Per-lane hit
// counters are summed across the 64-lane wave and the workgroup; one 64-bit atomic per workgroup.
// A pattern set over a small text runs as ONE grid, gridDim.y = pattern (launch_scan_set).
// Earlier designs of the serial kernels (so_scan, kmp_scan, so_runs64, so_runs1, kmp_links_runs,
// kmp_runs1) live in kernels_ab.inc / k_ab.hip and only in the A/B build (make AB=1, smartgpu_tune).
#include "launch_common.hpp"

#include "../../include/smartgpu.h"

namespace sg {

thread_local BatchCtx g_batch = {nullptr, 1};

int g_tune[8] = {0, 0, 0, 0, 0, 0, 0, 0};

// Which smartgpu_tune settings this build can honour: the kernels behind the others are only in the A/B build.
bool tune_supported(int key, int value)
{
#ifdef SMARTGPU_AB
    (void)key; (void)value;
    return true;
#else
    switch (key) {
        case 0: return value == 0 || value == 1 || value == 3;  // 2: Horspool on the bank-private tiles
        case 3: return value == 0 || value == 5;                 // superseded KMP kernels (5: kmp_runs without its four-byte table; 6: round 3's one-workgroup form)
        case 6: return value == 0 || value == 5;                 // superseded SO kernels (5: so_runs without the four-symbol table)
        case 7: return value == 0 || (value >= 5 && value <= 9);   // packed load policies (9: EPSM without its packed-symbol modes; 8: m <= 4 without v_mqsad)
        default: return true;
    }
#endif
}

// skip algorithms use the packed matcher up to this m (crossovers measured on 1 GiB rand128
// with non-temporal tile loads, profiles/r01): HOR/TUNEDBM/RAITA 7, BM 8, BNDM 11, QS 14, HASH3/5/8 32/64/28
static constexpr uint32_t packed_max_m(int algo)
{
    return (algo == SMARTGPU_HOR || algo == SMARTGPU_TUNEDBM || algo == SMARTGPU_RAITA) ? 7u
         : algo == SMARTGPU_BM ? 7u : algo == SMARTGPU_BNDM ? 10u  // re-measured with five workgroups per CU below 16 bytes (session u)
         : algo == SMARTGPU_QS ? 14u     // three LDS reads per window (text byte, next byte, table): later crossover
         : algo == SMARTGPU_HASH3 ? 24u  // q text reads + hash + table per window, shifts of at most m-q+1: the
         : algo == SMARTGPU_HASH5 ? 32u  //   tiles pass the packed matcher's 78-81 % only here (with four workgroups
         : algo == SMARTGPU_HASH8 ? 80u  //   per CU; HASH8 own/packed: m=32 63/79, 64: 78/81, 96: 82/81 — session u)
         : 0u;
}

uint32_t short_pattern_max_m(int algo)
{
    if (algo == SMARTGPU_SBNDM || algo == SMARTGPU_BNDML) return packed_max_m(SMARTGPU_BNDM);  // they share bndm_scan's crossover
    if (algo == SMARTGPU_KR) return 15;  // below 16 bytes only the low m bits of the rolled hash can be compared (launch_scan)
    return packed_max_m(algo);
}

// Regimes of the skip algorithms (HOR, BM, BNDM).  For short patterns the window is a
// few dwords and a skip loop degenerates (a lane advances ~m bytes per two dependent
// LDS reads); the packed matcher tests every alignment at HBM speed instead — the
// "hybrid" SURVEY.md §7 describes; counts are identical.  Measured on 1 GiB rand128
// (profiles/r01): HOR m=4: flat tile 45 %, bank-private 55 %, packed 74 % of 8 TB/s.
// Thresholds per algorithm: packed_max_m().
static int hor_regime(uint32_t m, int algo = SMARTGPU_HOR)
{
    const int v = g_tune[0];  // 0 auto, 1 flat, 2 bank-private, 3 packed
    if (v == 3) return 3;
    if (v == 2) return m <= kHaloMax + 1 ? 2 : 1;  // the bank-private kernel keeps whole windows in LDS
    if (v == 1) return 1;
    return m <= packed_max_m(algo) ? 3 : 1;  // bank-private kernel: only on request
}

const char* scan_kernel_name(int algo, uint32_t m, bool prefer_packed, bool so_masks, uint32_t halo)
{
#ifdef SMARTGPU_AB
    const char* const so_name = g_tune[6] == 1 ? "so_scan" : g_tune[6] == 2 ? "so_runs64" : g_tune[6] == 4 ? "so_runs1" : "so_runs";
    const char* const sa_name = (g_tune[6] == 3 || g_tune[6] == 4) ? "so_runs1" : "so_runs";
    const char* const kmp_name = (g_tune[3] == 1 && m <= 40) ? "kmp_scan" : g_tune[3] == 2 ? "kmp_links_runs" : g_tune[3] == 3 ? "kmp_runs1" : "kmp_runs";
    const char* const reroute_name = g_tune[6] == 4 ? "so_runs1" : "so_runs";
#else
    const char* const so_name = "so_runs";
    const char* const sa_name = "so_runs";
    const char* const kmp_name = "kmp_runs";
    const char* const reroute_name = "so_runs";
#endif
    if (so_masks && g_tune[0] == 0) return reroute_name;
    const bool pk = prefer_packed && g_tune[0] == 0;
    switch (algo) {
        case SMARTGPU_TUNEDBM:
        case SMARTGPU_HOR: {
            const int r = pk ? 3 : hor_regime(m);
            return r == 3 ? "packed_scan" : r == 2 ? "hor_scan_bp" : "hor_scan";
        }
        case SMARTGPU_HASH3:
        case SMARTGPU_HASH5:
        case SMARTGPU_HASH8:
        case SMARTGPU_RAITA:
        case SMARTGPU_QS: return (pk || hor_regime(m, algo) == 3) ? "packed_scan" : "hor_scan";
        case SMARTGPU_SA: return sa_name;
        case SMARTGPU_KR: return (m < 16 && g_tune[0] != 1) ? "packed_scan" : "hor_scan_bp";
        case SMARTGPU_BM: return (pk || m == 1 || (m <= packed_max_m(SMARTGPU_BM) && g_tune[0] != 1)) ? "packed_scan" : "bm_scan";
        case SMARTGPU_KMP: return kmp_name;
        case SMARTGPU_SO: return so_name;
        case SMARTGPU_BNDML:
            if (m > 32) return pk ? "packed_scan" : "bndml_scan";
            [[fallthrough]];
        case SMARTGPU_SBNDM:
        case SMARTGPU_BNDM:
            if (algo != SMARTGPU_SBNDM && (halo & kBndmGramWindow) && !pk) return "bndm_scan";  // a one-gram window (api.cpp)
            return (pk || (m <= packed_max_m(SMARTGPU_BNDM) && g_tune[0] != 1)) ? "packed_scan" : algo == SMARTGPU_SBNDM ? "sbndm_scan" : "bndm_scan";
        case SMARTGPU_EPSM: return "packed_scan";
    }
    return "?";
}

// What launch_scan fills in for the kernels before they run: where the packed matcher finds the fingerprint
// in this algorithm's blob, and the Shift-Or masks of SO / SA.  Applied to the by-value arguments of a single
// launch and, by the caller, to every element of a pattern set's argument array (launch_scan_set).
ScanArgs prepare_scan_args(int algo, ScanArgs a)
{
    const uint32_t m = a.m;
    switch (algo) {
        case SMARTGPU_KR: a.fp_off = kTableOff + 4; break;  // after the pattern's hash
        case SMARTGPU_BM: a.fp_off = kTableOff + ((1536 + 2 * (m + 1) + 3) & ~3u); break;  // after first, second, bc, gs, safe shift
        case SMARTGPU_BNDML:  // after the masks (W = 2) and the period / after B[256]
            a.fp_off = m > 32 ? kTableOff + 1024 * 2 + 4 : kTableOff + 1024;
            a.halo &= 0xFFu;
            break;
        case SMARTGPU_SBNDM:
        case SMARTGPU_BNDM:  // after B[256]; halo = bndm_scan's q, the plan's marks above it
            a.fp_off = kTableOff + 1024;
            a.halo &= 0xFFu;
            break;
        case SMARTGPU_EPSM: a.fp_off = kTableOff; break;
        case SMARTGPU_SO: a.so_off = kTableOff; break;
        case SMARTGPU_SA: a.so_off = g_tune[6] == 3 ? kTableOff + 1024 : kTableOff; break;
        case SMARTGPU_KMP: break;
        default:  // the Horspool family: the fingerprint after the u16 and u8 tables; halo = H (at most 32), Horspool's q above it
            a.fp_off = kTableOff + 768;
            a.halo &= 0xFFu;
            break;
    }
    return a;
}

hipError_t launch_scan(int algo, const ScanArgs& a_in, int num_cus, hipStream_t stream, TextCodes codes)
{
    if (a_in.s_end <= a_in.s_begin) return hipSuccess;
    const bool rerouted = a_in.so_off != 0 && algo != SMARTGPU_SO && algo != SMARTGPU_SA;  // plans of 2-3-symbol patterns (api.cpp)
    const ScanArgs a = prepare_scan_args(algo, a_in);
    const uint32_t m = a.m;
    // A pattern over two or three symbols, 16 bytes or longer: no byte, pair or dword of it tells a
    // window from its neighbours, so the skip kernels move one or two bytes at a time and the packed
    // matcher tests all four fingerprint dwords at every alignment (33-45 % on rand2, all of them).
    // The branch-free bit-parallel runs kernel does not care what the bytes are (66-70 %): plans of
    // such patterns carry Shift-Or masks as well (api.cpp build_blob) and count with it.
    if (rerouted && g_tune[0] == 0) {
#ifdef SMARTGPU_AB
        if (g_tune[6] == 4) {  // the rerouted patterns on so_runs1 too
            bool handled = false;
            const hipError_t e = launch_ab_so(SMARTGPU_HOR, a, num_cus, stream, &handled);
            if (handled) return e;
        }
#endif
        return launch_so_runs(a, false, num_cus, stream, codes);
    }
    const bool pk = a.prefer_packed && g_tune[0] == 0;
    switch (algo) {
        case SMARTGPU_TUNEDBM:  // hor_scan<.., 0> is Tuned BM's loop (see the kernel's comment)
        case SMARTGPU_HOR: {
            const int regime = pk ? 3 : hor_regime(m);
            if (regime == 3) return launch_packed(SMARTGPU_HOR, a, num_cus, stream);  // a.fp_off: prepare_scan_args
#ifdef SMARTGPU_AB
            if (regime == 2) return launch_hor_bp(a, num_cus, stream);
#endif
            // a TEXT of at most four byte values: Horspool on its grams (k_horg.hip) — eight one-bit symbols from 16 bytes on (two
            // values), four two-bit symbols from 8; a window shorter than two grams cannot be shifted by more than its own rule
            // allows (m - Q + 1).  tune(2, 4): never (A/B).  Any other text: the plan's q-gram (hash) table, or the byte table.
            if (g_tune[2] != 4) {  // (Tuned BM — tunedbm.c:38-62 — is Horspool's shift behind an unrolled skip loop: on grams the same kernel)
                const int gram = ((codes.one & 0xFFu) != 0xFFu && m >= 16) ? 1 : (codes.shift < 7 && m >= 8) ? 2 : 0;
                if (gram) return launch_hor_gram(a, gram, num_cus, stream);
            }
            return launch_hor(a, (a_in.halo >> 8) & 0xFFu, num_cus, stream);  // q: the plan's q-gram table (api.cpp), 0: the byte table
        }
        case SMARTGPU_KR:
            // Short patterns, like the skip algorithms': the packed matcher (72-77 %).  Only the low m bits
            // of the rolled hash can be compared, so one window end in 2^m is confirmed (m = 8: 32 %,
            // m = 12: 50 % of 8 TB/s), and below 8 the rolling form needs the outgoing byte (15 %).
            if (m < 16 && g_tune[0] != 1) return launch_packed(SMARTGPU_HOR, a, num_cus, stream);
            return launch_kr(a, num_cus, stream);
        case SMARTGPU_HASH3:
        case SMARTGPU_HASH5:
        case SMARTGPU_HASH8:
        case SMARTGPU_RAITA:
        case SMARTGPU_QS:  // the Horspool family on hor_scan's tiles; short patterns: packed regime as HOR
            if (pk || hor_regime(m, algo) == 3) return launch_packed(SMARTGPU_HOR, a, num_cus, stream);
            return launch_hor_var(algo, a, num_cus, stream);
        case SMARTGPU_BM:
            // m = 1 always: bm_scan reads the byte before the window's last along with it, and a one-byte
            // window has none (at a tile's first byte that read would leave the tile region)
            if (m == 1 || (m <= packed_max_m(SMARTGPU_BM) && g_tune[0] != 1) || pk) return launch_packed(SMARTGPU_BM, a, num_cus, stream);
            // a TEXT of at most four byte values: Boyer-Moore on its grams (k_bmg.hip), as Horspool above.  tune(2, 4): never (A/B)
            if (g_tune[2] != 4) {
                const int gram = ((codes.one & 0xFFu) != 0xFFu && m >= 16) ? 1 : (codes.shift < 7 && m >= 8) ? 2 : 0;
                if (gram) return launch_bm_gram(a, gram, num_cus, stream);
            }
            return launch_bm(a, num_cus, stream);
        case SMARTGPU_BNDML:
            if (m > 32) {  // multi-word vectors; m <= 32 is plain BNDM (bndml.c:44-75): falls through
                if (pk) return launch_packed(SMARTGPU_BNDM, a, num_cus, stream);
                return launch_bndml(a, num_cus, stream);
            }
            [[fallthrough]];
        case SMARTGPU_SBNDM:
        case SMARTGPU_BNDM:
            if (algo != SMARTGPU_SBNDM && (a_in.halo & kBndmGramWindow) && !pk) return launch_bndm(a, num_cus, stream, codes);  // a one-gram window
            if ((m <= packed_max_m(SMARTGPU_BNDM) && g_tune[0] != 1) || pk) return launch_packed(SMARTGPU_BNDM, a, num_cus, stream);
            if (algo == SMARTGPU_SBNDM) return launch_sbndm(a, num_cus, stream);
            return launch_bndm(a, num_cus, stream, codes);
        case SMARTGPU_SA:  // Shift-And counts in the complemented, Shift-Or form (api.cpp build_blob); its own AND form
        case SMARTGPU_SO: {  // (so_runs1<.., AND = true>, masks after the Shift-Or ones) is in the A/B build: tune(6,3)
#ifdef SMARTGPU_AB
            bool handled = false;
            const hipError_t e = launch_ab_so(algo, a, num_cus, stream, &handled);
            if (handled) return e;
#endif
            return launch_so_runs(a, false, num_cus, stream, codes);  // a.so_off: prepare_scan_args
        }
        case SMARTGPU_KMP: {
#ifdef SMARTGPU_AB
            bool handled = false;
            const hipError_t e = launch_ab_kmp(a, num_cus, stream, &handled);
            if (handled) return e;
#endif
            return launch_kmp_runs(a, num_cus, stream, codes);  // per-lane runs streamed through LDS
        }
        case SMARTGPU_EPSM: return launch_packed(SMARTGPU_EPSM, a, num_cus, stream, codes);  // a.fp_off: prepare_scan_args
    }
    return hipErrorInvalidValue;
}

hipError_t launch_scan_set(int algo, const ScanArgs& first, const BatchItem* device_items, uint32_t count, int num_cus,
                           hipStream_t stream, TextCodes codes)
{
    if (count == 0) return hipSuccess;
    g_batch = {device_items, count};
    const hipError_t e = launch_scan(algo, first, num_cus, stream, codes);
    g_batch = {nullptr, 1};
    return e;
}

}  // namespace sg
