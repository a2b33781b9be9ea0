// kernels.hpp — launch interface between the C-ABI layer (api.cpp) and the
// gfx950 scan kernels (kernels.hip).  Host-only types; no HIP headers needed
// by includers other than hipStream_t.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace sg {

// ---- text buffer geometry (HBM layout) --------------------------------------
// A text of n bytes lives in one allocation  [FRONT_PAD | n bytes | back pad],
// pads zero-filled.  Text byte 0 is 256-byte aligned.  The pads let every tile
// load (a tile plus its halo, rounded to 16 B) stay inside the allocation with
// no per-load bounds checks; pad bytes are never counted as text.
constexpr uint64_t kFrontPad = 4608 + 8704;     // >= kHitSlotsOff + 8 KB of staging slots + kXSize + 256, multiple of 256
constexpr uint32_t kHitSlots = 64;              // flush_hits (dev_common.hpp): staging slots for the workgroups' sums, 128 bytes apart,
constexpr uint32_t kHitSlotsOff = 512;          // at this offset of the text's allocation (below every byte a scan reads, above the alphabet scratch)
#ifndef SMARTGPU_HIT_SLOTS_MIN_GRID
#define SMARTGPU_HIT_SLOTS_MIN_GRID 1024
#endif
constexpr uint32_t kHitSlotsMinGrid = SMARTGPU_HIT_SLOTS_MIN_GRID;  // grids of fewer workgroups add to the result directly (variant builds: A/B)
constexpr uint64_t kBackPad = 160 * 1024;       // >= largest tile + kXSize + 64
constexpr uint32_t kBmHalo = 20;                // bm_scan's lane tiles: 16 bytes of the previous segment + 4 of padding per lane
constexpr uint32_t kHaloMax = 16;               // bytes a lane verifies by itself in LDS before it parks the window
constexpr uint32_t kPatternBytes = 4224;        // pattern slot in the plan blob (>= kXSize, /16)
constexpr int kResultSlots = 4096;
constexpr uint32_t kSoWindow = 29;     // SO/SA: bytes of the pattern so_runs keeps in its 32-bit state (four steps of the
                                       // recurrence at once need three bits of headroom); longer patterns: prefix + verification
constexpr uint32_t kKmpWindow = 254;   // KMP (kmp_runs): states 0..w plus the absorbing accept row are u8 ids: the whole pattern up to 254 bytes
// Longer patterns: the automaton of this prefix, hits verified.  A run re-scans w-1 bytes: with the 254-byte prefix of
// round 1 that was 6 % of a 4 KiB run; measured on 1 GiB, m = 256 .. 4096: 0.200-0.219 ms (254), 0.192-0.22 (126),
// 0.188-0.212 (62) on rand128, the same order on English and rand2.  62 keeps the ids 4s (tables.cpp).
constexpr uint32_t kKmpPrefix = 62;
constexpr uint32_t kmp_window(uint32_t m) { return m <= kKmpWindow ? m : kKmpPrefix; }
// kmp_runs<., false, COMPACT>: the automaton of the pattern or of its 56-byte prefix in a compact table — five four-wave
// workgroups per CU, each with its own copy (launch_common.hpp, k_kmp.hip)
#ifndef SMARTGPU_KMP_COMPACT_WINDOW
#define SMARTGPU_KMP_COMPACT_WINDOW 56  // (a macro so that tools/build_variant.sh can build other windows for an A/B)
#endif
constexpr uint32_t kKmpCompactWindow = SMARTGPU_KMP_COMPACT_WINDOW;  // 60 fits five workgroups into 160 KB on paper (162,640 bytes) and ran as four + a tail round: 0.24 ms where 0.19 was due
constexpr uint32_t kmp_compact_window(uint32_t m) { return m < kKmpCompactWindow ? m : kKmpCompactWindow; }
constexpr uint32_t kKmpQBytes = 272;   // kmp_runs: after the transitions, Q[s] = P[s..s+4) for 64 states (LDS), thr = 4K, 12 bytes of padding
constexpr uint32_t kKmpDfaMaxM = 255;  // KMP: the automaton's states are u8, so its (w+1)*256-byte transition
                                       // table (<= 64 KB of LDS) recognises w = min(m, 255) bytes; longer
                                       // patterns: the automaton of the 255-byte prefix + verification

// What every scan kernel receives.
struct ScanArgs {
    const uint8_t* text;        // device pointer to text byte 0
    uint64_t s_begin, s_end;    // start positions to count: s_begin <= s < s_end (s_end <= n-m+1)
    uint32_t m;                 // pattern length
    uint32_t halo;              // skip kernels: back-halo H = min(m-1, kHaloMax); serial: forward halo
    uint32_t fp_off;            // packed kernel: blob offset of the fingerprint (set by launch_scan)
    uint32_t prefer_packed;     // HOR/BM: the shift tables promise tiny shifts (small alphabet) -> packed regime
    uint32_t sparse;            // skip kernels: the pattern's own symbols promise long shifts and few candidates
                                // (api.cpp build_blob) -> fewer workgroups per CU, see kTileWgs
    uint32_t so_off;            // blob offset of Shift-Or masks u32 S[256]; set by launch_scan for SO/SA, by the plan
                                // for other algorithms when the pattern is best counted by so_runs (else 0)
    const uint8_t* blob;        // device: [pattern kPatternBytes][tables ...]
    unsigned long long* count;  // device result slot (pre-zeroed)
};

// What the TEXT consists of (api.cpp text_alphabet, taken once when the text is created): at most four distinct byte
// values whose bits shift, shift+1 tell them apart — shift < 7, symtab = the byte value of each two-bit code (unused
// codes: a byte the text does not hold) — or shift = 7.  The runs kernels then take four text bytes per table step
// (so_runs<., true>, kmp_runs<., true>).  The host picks the instantiation with this; the KERNELS read the two words
// from the first eight bytes of the text's own allocation (its front pad, which no load of a scan reaches: the pad is
// kXSize + 256 bytes and more).  Neither members of ScanArgs nor kernel arguments: two more words in either changed
// the register allocation and the schedule of kernels whose source had not changed — hor_scan ran 5-6 % slower with
// them in ScanArgs, kmp_runs<false, false> 3-9 % slower with them as its own arguments (build against build).
// `one` (round 4; the third word of the text's allocation): a text of at most TWO byte values also has one-bit codes —
// bits 0-7: the bit that tells the two apart (0xFF: more than two values), bits 8-15 / 16-23: the value whose bit is 0 / 1
// (bndm_scan<.., GRAM = 1>: eight symbols per table step).
struct TextCodes { uint32_t shift = 7, symtab = 0, one = 0xFF; };

// One pattern of a set that runs as ONE grid (launch_scan_set): what differs from pattern to pattern.  Everything
// else — text, range, m — and the BASES of the table arena and of the count array come from the by-value ScanArgs.
struct BatchItem {
    uint64_t blob_off;       // the pattern's blob inside the arena (ScanArgs.blob = arena base)
    uint32_t count_idx;      // its count slot (ScanArgs.count = first slot)
    uint32_t halo, fp_off, prefer_packed, sparse, so_off;  // as in ScanArgs, after prepare_scan_args
};

// Byte offsets of the tables inside the blob, after the pattern slot.
//  HOR : u16 tab[256]   shift | 0x8000 if c == P[m-1]; u8 tab8[256]; fingerprint (as EPSM)
//  BM  : u16 first[256] (last-byte shift | 0x8000 if c == P[m-1]), u16 second[256] (the same one byte earlier), u16 bc[256], u16 gs[m], u16 safe_shift
//  KMP : i16 next[m+1]; u8 dfa[256*256] of P[0..w), w = min(m, kKmpDfaMaxM), 16-byte aligned: state s is row
//        id(s) = rotl8(s,2) (accept state w: 255), entries are ids, row r XOR-swizzled: delta(r,c) at r*256 + (c ^ r)
//  SO  : u32 S[256]
//  BNDM: u32 B[256]
//  EPSM: u32 fp[4], u32 fpmask[4]   (first min(m,16) pattern bytes as dwords + byte masks)
constexpr uint32_t kTableOff = kPatternBytes;
// KMP: i16 next[m+1], then (16-byte aligned) the compact table of kmp_compact_window(m) — (w + 1) rows + Q —, then, where the
// plan carries one (a pattern over at most four symbols: the four-byte form; the A/B build: always), the spread table
constexpr uint32_t kmp_compact_off(uint32_t m) { return kTableOff + ((2 * (m + 1) + 15u) & ~15u); }
constexpr uint32_t kmp_spread_off(uint32_t m) { return kmp_compact_off(m) + (kmp_compact_window(m) + 1) * 256 + kKmpQBytes; }
constexpr uint32_t kBndmlWindow = 64;           // bytes of the pattern bndml_scan keeps in its bit vectors (multiple of 32, <= 256)

struct LaunchInfo {
    const char* kernel_name;  // as rocprofv3 prints it
};

// Enqueue the scan for `algo` on `stream`; returns hipSuccess or the launch error.
hipError_t launch_scan(int algo, const ScanArgs& a, int num_cus, hipStream_t stream, TextCodes codes = TextCodes());
ScanArgs prepare_scan_args(int algo, ScanArgs a);  // what launch_scan fills in (fp_off, so_off of SO/SA)
// The same for a pattern set in ONE grid (gridDim.y = count): `device_items` holds the per-pattern fields in
// device memory, `first` the common ones with blob = arena base and count = first count slot, plus the first
// pattern's plan fields (which choose kernel and grid: all patterns of the set must agree on prefer_packed,
// sparse and so_off != 0 — the caller groups them — and carry prepare_scan_args' fp_off / so_off).
hipError_t launch_scan_set(int algo, const ScanArgs& first, const BatchItem* device_items, uint32_t count, int num_cus,
                           hipStream_t stream, TextCodes codes = TextCodes());
// occurrence positions (extension): appends every s in [a.s_begin, a.s_end) with T[s..s+m) == P to `out`
// (unordered, at most `cap` entries), total in a.count; the blob must be an EPSM blob
hipError_t launch_find(const ScanArgs& a, unsigned long long* out, unsigned long long cap, int num_cus,
                       hipStream_t stream);
const char* scan_kernel_name(int algo, uint32_t m, bool prefer_packed, bool so_masks, uint32_t halo = 0);
// BNDM: bit 8 of the plan's halo (its low byte is bndm_scan's q) — a pattern of 8+ bytes over two to four symbols: on a text
// of at most four byte values bndm_scan<.., GRAM> decides every window with one lookup; bndm_scan at any length
constexpr uint32_t kBndmGramWindow = 0x100;
// longest pattern for which a skip algorithm's own LDS-tile loop is slower than an every-byte kernel (0: never)
uint32_t short_pattern_max_m(int algo);

// tuning knobs (smartgpu_tune): [0] HOR variant 0 auto / 1 flat / 2 bank-private
extern int g_tune[8];
bool tune_supported(int key, int value);  // false: the setting needs the A/B build (make AB=1)

// text generators / helpers (device side)
hipError_t launch_generate(uint8_t* dst, uint64_t seed, int sigma, uint64_t off, uint64_t n,
                           hipStream_t stream);
hipError_t launch_tile_fill(uint8_t* dst, const uint8_t* unit, uint64_t unit_len, uint64_t phase,
                            uint64_t n, hipStream_t stream);

// out[8] (device, zeroed by the caller): bit c set <=> byte value c occurs in text[0, n)
hipError_t launch_text_alphabet(const uint8_t* text, uint64_t n, uint32_t* out, int num_cus, hipStream_t stream);
hipError_t launch_probe_read(const uint8_t* text, uint64_t n, unsigned long long* sink, int num_cus,
                             hipStream_t stream);

}  // namespace sg
