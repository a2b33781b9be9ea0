// api.cpp — C ABI of libsmartgpu (include/smartgpu.h): text residency in HBM,
// host-side preprocessing, launches, timing.  Compiled with hipcc as host code.
//
// There is deliberately no CPU search path in this file or anywhere in the
// library: if HIP is not usable every compute entry point fails.
#include "../../include/smartgpu.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types, enums and prototypes only: the library itself is dlopen'ed on first use

#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "kernels.hpp"
#include "tables.hpp"

namespace {

thread_local std::string g_error;
thread_local double g_last_pre_ms = 0.0, g_last_run_ms = 0.0;

void set_error(const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_error = buf;
}

#define HIP_TRY(expr, fail)                                                          \
    do {                                                                             \
        hipError_t e_ = (expr);                                                      \
        if (e_ != hipSuccess) {                                                      \
            set_error("%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            fail;                                                                    \
        }                                                                            \
    } while (0)

constexpr int kMaxDevices = 16;

struct DeviceCtx {
    bool ready = false;
    int device = 0;
    hipStream_t stream = nullptr;
    int num_cus = 0;
    uint8_t* pinned = nullptr;      // staging for uploads
    size_t pinned_bytes = 0;
    unsigned long long* pinned_count = nullptr;  // 8-byte readback slot
    hipEvent_t mark[2] = {nullptr, nullptr};     // smartgpu_stream_mark()
    // smartgpu_search_batch64(): one arena for the K pattern blobs and the K counts, grown when a
    // batch needs more, never per pattern; K+1 events for the per-pattern device times
    uint8_t* arena = nullptr;
    size_t arena_bytes = 0;
    unsigned long long* batch_counts = nullptr;
    size_t batch_slots = 0;
    unsigned long long* pinned_counts = nullptr;  // host side of the one read-back
    std::vector<hipEvent_t> batch_events;
};
DeviceCtx g_dev[kMaxDevices];

DeviceCtx* device_ctx(int device)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        set_error("no HIP device available (hipGetDeviceCount)");
        return nullptr;
    }
    if (device < 0 || device >= n || device >= kMaxDevices) {
        set_error("device %d out of range (have %d)", device, n);
        return nullptr;
    }
    DeviceCtx& d = g_dev[device];
    HIP_TRY(hipSetDevice(device), return nullptr);
    if (!d.ready) {
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, device), return nullptr);
        d.num_cus = prop.multiProcessorCount;
        d.device = device;
        HIP_TRY(hipStreamCreateWithFlags(&d.stream, hipStreamNonBlocking), return nullptr);
        d.pinned_bytes = 32u << 20;
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&d.pinned), d.pinned_bytes, hipHostMallocDefault),
                return nullptr);
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&d.pinned_count), 64, hipHostMallocDefault),
                return nullptr);
        d.ready = true;
    }
    return &d;
}

// hor_flat, bm_scan, bndm_scan and kmp_runs address LDS by absolute offset; a kernel that finds its dynamic segment
// elsewhere than at offset 0 (a static __shared__ crept into it) adds 2^62 to its count and returns.  No text has
// 2^62 start positions: such a count is reported as an error, never as a result.
constexpr unsigned long long kPoisonedCount = 1ull << 62;
bool count_poisoned(unsigned long long c)
{
    if (c < kPoisonedCount) return false;
    set_error("a scan kernel found its LDS segment displaced (static LDS in a kernel that addresses LDS by offset): count poisoned");
    return true;
}

double now_ms()
{
    using clk = std::chrono::steady_clock;  // CLOCK_MONOTONIC, as src/timer.h:43-55
    return std::chrono::duration<double, std::milli>(clk::now().time_since_epoch()).count();
}

}  // namespace

struct smartgpu_text {
    int device = 0;
    uint64_t n = 0;
    uint8_t* base = nullptr;  // allocation start; text byte 0 at base + kFrontPad
    const uint8_t* data() const { return base + sg::kFrontPad; }
    // what the text consists of, taken once when it is created (text_alphabet below): the byte values that occur, and
    // — for at most four of them — their two-bit codes (ScanArgs.four_shift, four_symtab; 7: none)
    uint32_t alphabet[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t four_shift = 7, four_symtab = 0, one_bit = 0xFF;
    sg::TextCodes codes() const { sg::TextCodes c; c.shift = four_shift; c.symtab = four_symtab; c.one = one_bit; return c; }
};

struct smartgpu_plan {
    int device = 0;
    int algo = 0;
    uint32_t m = 0;
    uint32_t halo = 0;
    uint32_t prefer_packed = 0;  // see build_blob
    uint32_t sparse = 0;         // see build_blob
    uint32_t so_off = 0;         // see build_blob
    uint8_t* blob = nullptr;               // device: pattern + tables
    unsigned long long* results = nullptr; // device: kResultSlots counters (library-owned)
    unsigned long long* ext_results = nullptr; // caller-owned device buffer, if set
    int ext_slots = 0;
    unsigned long long* slot_ptr(int slot) const { return ext_results ? ext_results + slot : results + slot; }
    int num_slots() const { return ext_results ? ext_slots : sg::kResultSlots; }
    hipEvent_t ev0[sg::kResultSlots] = {}; // created lazily for timed launches
    hipEvent_t ev1[sg::kResultSlots] = {};
    bool timed[sg::kResultSlots] = {};
    double pre_ms = 0.0;
};

namespace {

const char* kAlgoNames[SMARTGPU_NUM_ALGOS] = {"hor", "bm", "kmp", "so", "bndm", "epsm", "sa", "qs", "tunedbm", "raita",
                                              "hash3", "hash5", "hash8", "sbndm", "kr", "bndml"};

// shortest pattern an algorithm applies to (the reference returns -1 below it: raita.c:37, hash3.c:31, ...)
uint32_t min_pattern(int algo)
{
    return (algo == SMARTGPU_RAITA || algo == SMARTGPU_SBNDM) ? 2u : algo == SMARTGPU_HASH3 ? 3u : algo == SMARTGPU_HASH5 ? 5u : algo == SMARTGPU_HASH8 ? 8u : 1u;
}

smartgpu_text* text_alloc(uint64_t n, int device, DeviceCtx** ctx_out)
{
    DeviceCtx* d = device_ctx(device);
    if (!d) return nullptr;
    smartgpu_text* t = new smartgpu_text;
    t->device = device;
    t->n = n;
    const uint64_t total = sg::kFrontPad + n + sg::kBackPad;
    if (hipMalloc(reinterpret_cast<void**>(&t->base), total) != hipSuccess) {
        set_error("hipMalloc of %llu bytes failed", (unsigned long long)total);
        delete t;
        return nullptr;
    }
    // pads are zero; they are never counted as text, only read by whole-tile loads
    if (hipMemsetAsync(t->base, 0, sg::kFrontPad, d->stream) != hipSuccess ||
        hipMemsetAsync(t->base + sg::kFrontPad + n, 0, sg::kBackPad, d->stream) != hipSuccess) {
        set_error("hipMemsetAsync of the text pads failed");
        hipFree(t->base);
        delete t;
        return nullptr;
    }
    *ctx_out = d;
    return t;
}

// One pass over a text that has just been written (upload, tile fill, generator): which byte values it holds.  A text
// is never written again (ownership), so the answer holds for every later search, of any part of it.
// The 256 bits collect in the text's OWN front pad (bytes 64..95 of the allocation — zeroed with the pad; no scan reads
// below byte 152 of it: the pad is kXSize + 256 bytes and more), not in a per-device buffer: two threads that create
// texts on one device cannot interleave their passes.
constexpr size_t kAlphabetScratchOff = 64;
static_assert(kAlphabetScratchOff + 32 <= sg::kFrontPad - (SMARTGPU_XSIZE + 256), "the scratch must lie below every byte a scan can read");
static_assert(sg::kHitSlotsOff >= kAlphabetScratchOff + 32 && sg::kHitSlotsOff + sg::kHitSlots * 128 <= sg::kFrontPad - (SMARTGPU_XSIZE + 256),
              "flush_hits' staging slots lie between the scratch and the bytes a scan can read");
bool text_alphabet(smartgpu_text* t, DeviceCtx* d)
{
    uint32_t* dev = reinterpret_cast<uint32_t*>(t->base + kAlphabetScratchOff);
    hipError_t e = hipMemsetAsync(dev, 0, 32, d->stream);
    if (e == hipSuccess && t->n) e = sg::launch_text_alphabet(t->data(), t->n, dev, d->num_cus, d->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(t->alphabet, dev, 32, hipMemcpyDeviceToHost, d->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(d->stream);
    if (e != hipSuccess) { set_error("text_alphabet failed: %s", hipGetErrorString(e)); return false; }
    if (!sg::four_symbol_codes(t->alphabet, &t->four_shift, &t->four_symtab)) t->four_shift = 7;
    // one-bit codes of a text of at most two byte values: the lowest bit in which they differ (one value: bit 0)
    t->one_bit = 0xFF;
    {
        uint32_t syms[3], ns = 0;
        for (uint32_t c = 0; c < 256 && ns < 3; ++c)
            if (t->alphabet[c >> 5] >> (c & 31) & 1u) syms[ns++] = c;
        if (ns == 1) syms[ns++] = syms[0];
        if (ns == 2) {
            uint32_t bit = 0;
            while (bit < 7 && ((syms[0] ^ syms[1]) >> bit & 1u) == 0) ++bit;
            const uint32_t s0 = (syms[0] >> bit & 1u) ? syms[1] : syms[0], s1 = (syms[0] >> bit & 1u) ? syms[0] : syms[1];
            t->one_bit = bit | (s0 << 8) | (s1 << 16);
        }
    }
    // the kernels' copy: the first three words of the allocation (TextCodes, kernels.hpp)
    const uint32_t words[3] = {t->four_shift, t->four_symtab, t->one_bit};
    if (hipMemcpyAsync(t->base, words, 12, hipMemcpyHostToDevice, d->stream) != hipSuccess ||
        hipMemsetAsync(dev, 0, 32, d->stream) != hipSuccess ||  // the pad is zero again
        hipStreamSynchronize(d->stream) != hipSuccess) {
        set_error("text_alphabet: writing the codes failed");
        return false;
    }
    return true;
}

// A text that is searched ONCE (the int search(P, m, T, n) shims upload it per call): no alphabet pass — a full
// read of the text and two synchronisations for a table form that one search does not pay back.  The runs kernels
// then take their byte-wise forms (shift 7 = "not a four-symbol text"), which are correct on any text.
bool text_no_alphabet(smartgpu_text* t, DeviceCtx* d)
{
    t->four_shift = 7;
    t->four_symtab = 0;
    t->one_bit = 0xFF;
    for (uint32_t& wv : t->alphabet) wv = 0xFFFFFFFFu;  // unknown: every byte value may occur
    const uint32_t words[3] = {7u, 0u, 0xFFu};
    if (hipMemcpyAsync(t->base, words, 12, hipMemcpyHostToDevice, d->stream) != hipSuccess || hipStreamSynchronize(d->stream) != hipSuccess) {
        set_error("text upload: writing the codes failed");
        return false;
    }
    return true;
}

// Build the device blob (pattern + tables) for (algo, P, m) in `blob` (cleared first; a caller that builds many
// reuses one vector: fresh memory for every blob of a pattern set cost more in page faults than the tables in work).
void build_blob(std::vector<uint8_t>& blob, int algo, const uint8_t* P, uint32_t m, uint32_t* halo,
                uint32_t* prefer_packed, uint32_t* sparse, uint32_t* so_off)
{
    *prefer_packed = 0;
    *sparse = 0;
    *so_off = 0;
    blob.clear();
    blob.resize(sg::kPatternBytes, 0);
    std::memcpy(blob.data(), P, m);
    auto append = [&blob](const void* p, size_t bytes) {
        const uint8_t* b = static_cast<const uint8_t*>(p);
        blob.insert(blob.end(), b, b + bytes);
    };
    // first min(m,16) pattern bytes as 4 dwords + 4 byte masks (packed kernel)
    auto append_fingerprint = [&]() {
        uint32_t fp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        uint8_t* fb = reinterpret_cast<uint8_t*>(fp);
        const uint32_t F = std::min<uint32_t>(m, 16);
        for (uint32_t i = 0; i < F; ++i) {
            fb[i] = P[i];
            fb[16 + i] = 0xFF;
        }
        append(fp, sizeof fp);
    };
    // How often two text symbols are equal, if the text's symbols are distributed like the pattern's
    // own (unbiased from its histogram: sum c(c-1) / m(m-1)).  Above 1/48 — binary and DNA-like
    // alphabets, natural language (~1/15), rand32 — a skip loop meets a candidate every few windows,
    // its lanes verify and diverge (English, m >= 64: BM 31-44 %, BNDM 31-36 %, HOR 63-69 %; rand32:
    // 67-75 %), and the packed matcher, which does the same work whatever the bytes are, is the better
    // regime (64-81 % on all of them).  Below it (rand64 and up) windows die on their first comparison
    // and the skip kernels only stream (80-86 % on rand128 against 76-81 %).  Measured: DESIGN.md §8.
    // The count does not depend on the choice.
    bool repeats = false;
    bool repeats_short = false;  // 7 bytes or fewer with a symbol that occurs twice (a small alphabet, most likely)
    uint64_t pairs = 0;          // ordered pairs of equal symbols in P: pairs / (m (m-1)) estimates P(two symbols are equal)
    uint32_t distinct = 0;       // symbols that occur in P
    uint32_t bndm_q_wanted = 1;  // BNDM: the q its statistics ask for (the plan's q also has to divide the window)
    uint32_t hor_q = 0;          // HOR: the q of its q-gram bad-character table (patterns over two to four symbols), 0: the byte table
    {
        uint32_t cnt[256] = {0};
        for (uint32_t i = 0; i < m; ++i) distinct += cnt[P[i]]++ == 0;
        for (uint32_t c = 0; c < 256; ++c) pairs += static_cast<uint64_t>(cnt[c]) * (cnt[c] > 0 ? cnt[c] - 1 : 0);
        // m < 32: too few symbols for the estimate; two equal pairs are taken as a sign (rand32, m = 16:
        // skip kernels 58-65 %, packed 76 %; on rand128 one pattern in four then goes packed, 76 % for 83 %)
        repeats = m > 7 && (pairs * 48 > static_cast<uint64_t>(m) * (m - 1) || (m < 32 && pairs >= 4));
        repeats_short = m <= 7 && pairs >= 2;
    }
    // The opposite case: symbols do not repeat (random text over a large alphabet).  Windows then die on
    // their first comparison and a skip kernel only streams; it runs best with FEWER workgroups per CU
    // (kTileWgs in kernels.hip, measured).  Patterns with repeating symbols that still run on a tile
    // kernel (tune(0,1), KR) keep the lanes busy verifying: more workgroups hide that.  How many exactly
    // depends on m and the kernel: tile_wgs() in kernels.hip.
    // (An earlier form — >= 80 % of the first min(m,64) symbols distinct — missed rand128 at m = 64,
    // 50 distinct symbols expected: HOR 80 %, BM 76 %, BNDM 71 % there against 84-87 % at m = 32 and 128.)
    *sparse = m >= 8 && !repeats;
    *halo = std::min<uint32_t>(m - 1, sg::kHaloMax);
    switch (algo) {
        case SMARTGPU_TUNEDBM:  // tunedbm.c:38-40: the same table with a zero for P[m-1] — the flag bit below
        case SMARTGPU_RAITA:    // raita.c:43: the same table
        case SMARTGPU_HOR: {
            const std::vector<int32_t> bc = sg::bad_char(P, m);
            std::vector<uint16_t> tab(256);
            for (int c = 0; c < 256; ++c)
                tab[c] = static_cast<uint16_t>(bc[c]) | (c == P[m - 1] ? 0x8000u : 0u);
            // Horspool over a pattern of two to four symbols (binary texts, DNA): the last BYTE says next to nothing — the
            // shift is the distance to its previous occurrence, a byte or two (hor_scan on rand2: 16 % of 8 TB/s at any m).
            // The bad-character rule on the window's last q BYTES instead (Horspool on the super-alphabet of q-grams, the
            // table indexed by their 8-bit hash sum T[e-k] 2^k — on a 0/1 text the q-gram itself — as Lecroq's hash3/5/8.c
            // do it): the shift is back at m - q + 1 for a gram that does not occur in P.  q = 8 for two symbols, 5 or 8 for
            // three and four; the kernel is hor_scan's HASHq instantiation, the table has Horspool's layout (zero entry =
            // flag | shift after a candidate).  Measured on 1 GiB, own kernel: rand2 m = 32 / 64 / 256+: 54 / 68 / 70 %
            // (byte table: 16 %), rand4 m = 16 / 32 / 64+: 60 / 73 / 74 % (45-50 %).  q travels in bits 8.. of the plan's halo.
            if (algo == SMARTGPU_HOR && distinct >= 2 && distinct <= 4 && m >= 16) {
                hor_q = distinct <= 2 ? 8u : m < 48 ? 5u : 8u;
                int32_t after = 1;
                const std::vector<int32_t> sh = sg::qgram_hash_shifts(P, m, hor_q, &after);
                for (int c = 0; c < 256; ++c)
                    tab[c] = sh[c] == 0 ? static_cast<uint16_t>(0x8000u | after) : static_cast<uint16_t>(sh[c]);
                *halo |= hor_q << 8;
            }
            append(tab.data(), 512);
            uint8_t tab8[256];  // plain u8 shifts for the bank-private kernel (m <= 255)
            for (int c = 0; c < 256; ++c) tab8[c] = static_cast<uint8_t>(bc[c] > 255 ? 255 : bc[c]);
            append(tab8, 256);
            append_fingerprint();  // packed regime
            *prefer_packed = repeats;
            break;
        }
        case SMARTGPU_BM: {
            const std::vector<int32_t> bc = sg::bad_char(P, m);
            const std::vector<int32_t> gs = sg::good_suffix(P, m);
            std::vector<uint16_t> tab(768 + m + 1);
            for (int c = 0; c < 256; ++c) {
                // shift after a mismatch on the window's last byte (bm.c:89 with i = m-1), or the
                // "last byte matches" flag
                tab[c] = c == P[m - 1] ? 0x8000u : static_cast<uint16_t>(std::max(gs[m - 1], bc[c]));
                // the same for the byte before it (bm.c:89 with i = m-2: max(bmGs[m-2], bmBc[c] - 1)):
                // most windows that survive their last byte die here, and the kernel has this byte
                // in hand already.  m = 1 has no such byte: all flags, the walk finds nothing to do.
                tab[256 + c] = (m < 2 || c == P[m - 2]) ? 0x8000u
                                                        : static_cast<uint16_t>(std::max(gs[m - 2], bc[c] - 1));
                tab[512 + c] = static_cast<uint16_t>(bc[c]);
            }
            for (uint32_t i = 0; i < m; ++i) tab[768 + i] = static_cast<uint16_t>(gs[i]);
            // gs[m]: a shift that is safe for a window whose last H+1 bytes matched, whatever
            // the remaining comparison says: min of gs over the unchecked positions (and gs[0],
            // the shift after a full match).  Used when the kernel parks the window.
            int32_t safe = gs[0];
            if (m - 1 > *halo)
                for (uint32_t i = 0; i + 1 < m - *halo; ++i) safe = std::min(safe, gs[i]);
            tab[768 + m] = static_cast<uint16_t>(safe);
            append(tab.data(), tab.size() * 2);
            if (blob.size() % 4) blob.resize((blob.size() + 3) & ~size_t(3), 0);
            append_fingerprint();  // packed regime
            *prefer_packed = repeats;
            break;
        }
        case SMARTGPU_KMP: {
            const std::vector<int32_t> nx = sg::kmp_next(P, m);
            std::vector<int16_t> tab(m + 1);
            for (uint32_t i = 0; i <= m; ++i) tab[i] = static_cast<int16_t>(nx[i]);
            append(tab.data(), tab.size() * 2);
            *halo = m - 1;  // forward halo: the automaton re-scans m-1 bytes
            blob.resize((blob.size() + 15) & ~size_t(15), 0);  // the transition table is 16-byte aligned
            // kmp_runs<., false, COMPACT> (round 4; every text that is not a four-symbol text): the automaton over
            // kmp_compact_window(m) bytes — the pattern, or its 56-byte prefix — with an ABSORBING accept row Z (every transition
            // into the accept state w leads to Z, Z leads to Z; row id(w) holds the real delta(w, .)) and the table of the
            // four-bytes-at-a-time forms (tables.cpp), stored as the kernel keeps it: row s at s * 256.
            sg::kmp_runs_tables(P, sg::kmp_compact_window(m), blob, true);
            // A pattern over at most four symbols (DNA-like alphabets): on a text that itself holds at most four byte values
            // kmp_runs takes FOUR text bytes per table step (kmp_runs<., FOUR>; that table lives in the gaps of the SPREAD
            // byte table, whose ids are 4s up to 62 states, and is derived from it on the device) — the window is then the
            // pattern or its 62-byte prefix; its length travels as the plan's prefer_packed field (0: no such table).  Which
            // of the two kernels runs is decided at launch, from the TEXT (TextCodes): each finds its own table.
            const uint32_t w4 = std::min<uint32_t>(m, sg::kKmpPrefix);
            uint32_t distinct4 = 0;
            {
                bool have[256] = {false};
                for (uint32_t i = 0; i < w4; ++i)
                    if (!have[P[i]]) { have[P[i]] = true; ++distinct4; }
            }
            const bool has4 = m >= 2 && distinct4 <= 4;
#ifdef SMARTGPU_AB
            sg::kmp_runs_tables(P, has4 ? w4 : sg::kmp_window(m), blob);  // the A/B build: always (tune(3,6): round 3's one-workgroup form)
#else
            if (has4) sg::kmp_runs_tables(P, w4, blob);
#endif
            if (has4) *prefer_packed = w4;
#ifdef SMARTGPU_AB
            {   // kmp_runs1 (A/B build): the automaton of P[0..w), w = min(m, 255); state s is row id(s) = rotl8(s, 2),
                // the accept state row 255 — or row 4w while the ids 4s do not wrap (w < 64) —, the largest id (its
                // running maximum needs that); row r XOR-swizzled by r
                const uint32_t w = std::min<uint32_t>(m, sg::kKmpDfaMaxM);
                const std::vector<uint8_t> dfa = sg::kmp_dfa(P, w);
                const uint32_t acc = w < 64 ? 4 * w : 255u;
                auto id = [w, acc](uint32_t st) { return st == w ? acc : ((st << 2) | (st >> 6)) & 255u; };
                std::vector<uint8_t> sw(256 * 256, 0);
                for (uint32_t st = 0; st <= w; ++st) {
                    const uint32_t r = id(st);
                    for (uint32_t c = 0; c < 256; ++c) sw[r * 256 + (c ^ r)] = static_cast<uint8_t>(id(dfa[st * 256 + c]));
                }
                append(sw.data(), sw.size());
            }
#endif
            break;
        }
        case SMARTGPU_SO: {
            const std::vector<uint32_t> S = sg::shift_or_masks(P, m);
            append(S.data(), 1024);
            break;
        }
        case SMARTGPU_BNDML:
            if (m > 32) {  // bndml.c:91-93: bit i of word i/32 of B[c] <=> P[w-1-i] == c, over w = min(m, kBndmlWindow) bytes
                const uint32_t w = std::min<uint32_t>(m, sg::kBndmlWindow);
                const uint32_t W = w <= 64 ? 2 : w <= 128 ? 4 : 8;
                std::vector<uint32_t> B(256 * W, 0u);
                for (uint32_t i = 0; i < w; ++i) B[P[w - 1 - i] * W + i / 32] |= 1u << (i % 32);
                append(B.data(), B.size() * 4);
                const std::vector<int32_t> nx = sg::kmp_next(P, w);  // nx[w] = longest proper border of P[0..w)
                const uint32_t period = w - static_cast<uint32_t>(nx[w]);
                append(&period, 4);
                append_fingerprint();  // packed regime (at kTableOff + 1024*W + 4)
                *prefer_packed = repeats;
                break;
            }
            [[fallthrough]];  // m <= 32: plain BNDM (bndml.c:44-75)
        case SMARTGPU_SBNDM:
        case SMARTGPU_BNDM: {
            const std::vector<uint32_t> B = sg::bndm_masks(P, m);
            append(B.data(), 1024);
            append_fingerprint();  // packed regime
            if (algo != SMARTGPU_SBNDM) {
                // bndm_scan reads q bytes of a window per iteration: the smallest q of 1, 2, 4, 8 (at most half the
                // window, and a divisor of it) for which a window rarely outlives its first iteration — the chance that the q-gram at its
                // end occurs in P[0..w), (w-q+1) * p^q with p = P(two symbols are equal) estimated from the pattern
                // itself, is below 0.3.  rand128: 1; English: 2; four symbols: 4; two: 8.  Travels as the plan's halo
                // (bndm_scan's lane tiles always hold the 32 bytes before a segment).
                const uint32_t w = std::min<uint32_t>(m, 32);
                double p = m > 1 ? std::max(static_cast<double>(pairs) / (static_cast<double>(m) * (m - 1)), 1.0 / 256) : 1.0;
                // (few symbols for the estimate: a pattern in which at most half of the symbols are distinct comes from
                // an alphabet of about that many — rand4, m = 8: the pair count says 0.05 .. 0.4, four symbols say 0.25)
                if (distinct * 2 <= m) p = std::max(p, 1.0 / distinct);
                uint32_t q = 1;
                while (q < 8 && 2 * q <= std::max(w / 2, 1u)) {
                    double survive = static_cast<double>(w - q + 1);
                    for (uint32_t i = 0; i < q; ++i) survive *= p;
                    if (survive < 0.3) break;
                    q *= 2;
                }
                bndm_q_wanted = q;
                while (w % q) q /= 2;  // q | w: a window is read through in whole iterations
                *halo = q;
            }
            if (algo == SMARTGPU_SBNDM) {  // sbndm.c:44-55: the shift after an occurrence = period of P[0..w)
                const uint32_t w = std::min<uint32_t>(m, 32);
                const std::vector<int32_t> nx = sg::kmp_next(P, w);  // nx[w] = longest proper border of the prefix
                const uint32_t period = w - static_cast<uint32_t>(nx[w]);
                append(&period, 4);
            }
            *prefer_packed = repeats;
            break;
        }
        case SMARTGPU_EPSM:
            append_fingerprint();
            break;
        case SMARTGPU_KR: {  // kr.c:38-41: the pattern's hash; weights 2^(m-1-i) mod 2^32 vanish beyond 32 bytes
            uint32_t hp = 0;
            for (uint32_t i = 0; i < m; ++i) hp = (hp << 1) + P[i];
            append(&hp, 4);
            append_fingerprint();  // packed regime (m < 16), at kTableOff + 4
            *halo = std::min<uint32_t>(m - 1, 32);  // bytes confirmed in LDS (and the hash's reach) behind a window end
            break;
        }
        case SMARTGPU_SA: {
            // sa.c:52 is D = ((D << 1) | 1) & S[c]; with E = ~D that is E = (E << 1) | ~S[c] — Shift-Or's step
            // on Shift-Or's masks, and "bit w-1 of D set" is "bit w-1 of E clear": the same automaton in
            // complement.  The kernel runs that form (one VALU op per byte less: there is no shift-and-AND
            // instruction); the AND form stays selectable (smartgpu_tune(6,3)) on the masks that follow.
            const std::vector<uint32_t> So = sg::shift_or_masks(P, m);
            append(So.data(), 1024);
            const std::vector<uint32_t> S = sg::shift_and_masks(P, m);
            append(S.data(), 1024);
            break;
        }
        case SMARTGPU_HASH3:
        case SMARTGPU_HASH5:
        case SMARTGPU_HASH8: {  // same layout as HOR: u16 table (zero entry = flag | shift after a candidate)
            const uint32_t q = algo == SMARTGPU_HASH3 ? 3 : algo == SMARTGPU_HASH5 ? 5 : 8;
            int32_t after = 1;
            const std::vector<int32_t> sh = sg::qgram_hash_shifts(P, m, q, &after);
            std::vector<uint16_t> tab(256);
            for (int c = 0; c < 256; ++c)
                tab[c] = sh[c] == 0 ? static_cast<uint16_t>(0x8000u | after) : static_cast<uint16_t>(sh[c]);
            append(tab.data(), 512);
            const uint8_t spare[256] = {0};
            append(spare, 256);
            append_fingerprint();  // packed regime
            *prefer_packed = repeats;
            break;
        }
        case SMARTGPU_QS: {  // same layout as HOR: u16 table, 256 spare bytes, fingerprint
            const std::vector<int32_t> qs = sg::quick_search_shifts(P, m);
            std::vector<uint16_t> tab(256);
            for (int c = 0; c < 256; ++c) tab[c] = static_cast<uint16_t>(qs[c]);
            append(tab.data(), 512);
            const uint8_t spare[256] = {0};
            append(spare, 256);
            append_fingerprint();  // packed regime
            *prefer_packed = repeats;
            break;
        }
    }
    // Patterns that are counted by the Shift-Or runs kernel whatever the algorithm (kernels.hip launch_scan): their
    // plans carry its masks as well.
    //  * Symbols repeat (the rule above; or, for 7 bytes and fewer, some symbol occurs twice): small alphabets,
    //    natural language, rand32.  Round 1 sent these to the packed matcher, which tests two or three fingerprint
    //    dwords in most rows there (rand4 64-72 %, English below 32 bytes 70-73 %, rand32 short 72-76 %); so_runs
    //    does the same work whatever the bytes are and since round 2 runs at 75-81 % on all of them (English from 32
    //    bytes on: 79-80 % both).  On random text over a large alphabet symbols do not repeat and nothing changes.
    //  * 8+ bytes whose first dword says next to nothing about where the pattern occurs (two to four symbols:
    //    every lane of the packed matcher keeps a candidate through all four fingerprint dwords, every skip is a byte
    //    or two): the chance that 4 text bytes drawn like the pattern's own equal P[0..4), times the 16 alignments a
    //    lane tests, is 0.03 or more (8+ bytes; up to four symbols).
    // SO and SA keep their own serial kernel, KMP its own from 9 bytes on — where its automaton can have the borderless
    // states 0..4 that the four-bytes-at-a-time form for natural language and medium alphabets needs (kmp_runs: K + 4 < w).
    // Below that kmp_runs has its state-0 form from 5 bytes and a lookup per byte otherwise, and counts frequent
    // occurrences byte by byte: m = 8: 61-72 % against so_runs' 74-81 % on every corpus; m = 2: 54 % against 76-78 %.
    // Karp-Rabin its own from 16 bytes on, EPSM its packed matcher (it IS that algorithm).
    if ((algo != SMARTGPU_KMP || m < 9) && algo != SMARTGPU_SO && algo != SMARTGPU_SA && (algo != SMARTGPU_KR || m < 16)) {
        // * Short patterns (below the algorithm's measured crossover with its own skip loop, kernels.hip packed_max_m):
        //   the every-byte kernels win there; so_runs and the packed matcher are equal on rand128 (76-77 %), so_runs
        //   ahead on everything else (rand256, rand32, English at m = 2, 4: 78-81 % against 67-76 %).
        bool to_so = (algo == SMARTGPU_KR || algo == SMARTGPU_KMP) ? true : (repeats || repeats_short || m <= sg::short_pattern_max_m(algo)) && algo != SMARTGPU_EPSM;
        // Round 3: where the algorithm's OWN kernel holds on such patterns, it keeps them (a configuration that
        // names Boyer-Moore or BNDM should not be a Shift-Or measurement).  bm_scan's flat loop: 72-74 % on English from 8
        // bytes on (so_runs 79-81 %); an alphabet of a few symbols shifts it by a byte or two (16-60 %): stays with so_runs.
        // bndm_scan with q-grams: 74-79 % on four symbols and 75-76 % on English from 16 bytes on, 66-71 % on two from 32
        // (q = 8); below those lengths it is iteration-bound.
        bool own_holds = false;
        // (hor_scan's flat form, VAR 9, for Horspool and Tuned BM: 68-70 % on English and rand32 from 8 bytes on)
        if (algo == SMARTGPU_BM || algo == SMARTGPU_HOR || algo == SMARTGPU_TUNEDBM)
            own_holds = m >= 8 && !(distinct <= 8 && 2 * distinct <= m);  // not: a few symbols, each several times
        // ... unless Horspool has its q-gram table for them and a window long enough for its shifts (within 5-10 points of so_runs)
        if (hor_q) own_holds = distinct <= 2 ? m >= 64 : m >= 32;
        // ... and, round 4, Horspool on GRAMS (k_horg.hip: one exact gram lookup per window on a text of at most four byte values,
        // 0.7-0.8 of the roofline from 8 bytes on): like BNDM's gram form a bet that the text holds no more byte values than
        // the pattern (on any other text the launch falls back to the hash table above, or the byte table)
        // (own kernel, 1 GiB: two values m = 16 / 32 / 64+: 0.66 / 0.76 / 0.79-0.83, four values: 0.75 / 0.79 / 0.80-0.85; 8 bytes:
        // 0.37 / 0.63 — a window of one or two grams: those stay with so_runs)
        // (Boyer-Moore likewise, k_bmg.hip: its good-suffix rule joined with the gram's shift)
        // (two symbols, 16 bytes — a window of two grams —: 0.66-0.69 against so_runs' 0.77: those from 32 bytes on, 0.78-0.81)
        if ((algo == SMARTGPU_HOR || algo == SMARTGPU_BM || algo == SMARTGPU_TUNEDBM) && distinct >= 2 && distinct <= 4) own_holds = m >= (distinct <= 2 ? 32u : 16u);
        if (algo == SMARTGPU_BNDM || (algo == SMARTGPU_BNDML && m <= 32)) own_holds = *halo == bndm_q_wanted && (*halo >= 8 ? m >= 32 : m >= 16);  // *halo: bndm_scan's q
        // BNDM over two to four symbols, 8+ bytes: on a text of at most four byte values bndm_scan's GRAM form decides every
        // window with one lookup (k_bndm.hip bndm_gram: 0.74-0.8 of the roofline on rand2 / rand4 at any such length; so_runs
        // 0.77).  The plan cannot see the text; a pattern of two to four symbols cut from it says what it most likely is
        // (on any other text the launch falls back to the mask loop — correct, and slow on such a pattern).
        const bool gram_window = (algo == SMARTGPU_BNDM || (algo == SMARTGPU_BNDML && m <= 32)) && m >= 8 && distinct >= 2 && distinct <= 4;
        if (gram_window) {
            *halo |= sg::kBndmGramWindow;
            to_so = false;
            *prefer_packed = 0;
        } else if (own_holds && m > sg::short_pattern_max_m(algo)) {
            to_so = false;
            *prefer_packed = 0;
        } else if (!to_so && m >= 8) {  // (8 bytes of distinct symbols estimate 16/8^4 = 0.004: below that the histogram says nothing)
            uint32_t cnt[256] = {0};
            for (uint32_t i = 0; i < m; ++i) ++cnt[P[i]];
            double pass = 16.0;
            for (uint32_t i = 0; i < 4; ++i) pass *= static_cast<double>(cnt[P[i]]) / m;
            // rand2: 1.0, rand3: 0.2, rand4: 0.06, rand8: 0.004.  Round 1 drew the line at 0.15 (so_runs then ran at
            // 63-68 %); at 75-81 % it also beats the packed matcher on rand4 (EPSM there: 59-67 %, a second
            // fingerprint dword in nearly every row), not on rand8 and up.
            to_so = pass >= 0.03;
            // EPSM keeps every pattern but the long ones over two symbols: its v_mqsad modes do the same work whatever the text holds
            // (k_packed.hip: 0.78-0.80 of the roofline on four byte values at any length; two values from 13 bytes on: 0.70-0.74, so_runs 0.77)
            if (algo == SMARTGPU_EPSM) to_so = distinct <= 2 && m >= 13;
        }
        if (to_so) {
            blob.resize((blob.size() + 15) & ~size_t(15), 0);
            *so_off = static_cast<uint32_t>(blob.size());
            const std::vector<uint32_t> S = sg::shift_or_masks(P, m);
            append(S.data(), 1024);
        }
    }
    blob.resize((blob.size() + 255) & ~size_t(255), 0);
}

std::vector<uint8_t> build_blob(int algo, const uint8_t* P, uint32_t m, uint32_t* halo,
                                uint32_t* prefer_packed, uint32_t* sparse, uint32_t* so_off)
{
    std::vector<uint8_t> blob;
    build_blob(blob, algo, P, m, halo, prefer_packed, sparse, so_off);
    return blob;
}

int check_search_args(int algo, const uint8_t* P, uint32_t m, const smartgpu_text* text,
                      uint64_t off, uint64_t n)
{
    if (algo < 0 || algo >= SMARTGPU_NUM_ALGOS) { set_error("unknown algorithm id %d", algo); return SMARTGPU_ERR_ARG; }
    if (m < min_pattern(algo)) { set_error("%s: not applicable for m < %u", kAlgoNames[algo], min_pattern(algo)); return SMARTGPU_NA; }
    if (!P || m < 1 || m > SMARTGPU_XSIZE) { set_error("pattern length %u outside [1,%d]", m, SMARTGPU_XSIZE); return SMARTGPU_ERR_ARG; }
    if (!text) { set_error("text handle is NULL"); return SMARTGPU_ERR_ARG; }
    if (off > text->n || n > text->n - off) { set_error("range [%llu,+%llu) outside the text (%llu bytes)", (unsigned long long)off, (unsigned long long)n, (unsigned long long)text->n); return SMARTGPU_ERR_ARG; }
    return SMARTGPU_OK;
}

sg::ScanArgs make_args(const smartgpu_plan* p, const smartgpu_text* text, uint64_t off, uint64_t n,
                       int slot)
{
    sg::ScanArgs a;
    a.text = text->data();
    a.s_begin = off;
    a.s_end = (n >= p->m) ? off + n - p->m + 1 : off;  // no window fits: empty range
    a.m = p->m;
    a.halo = p->halo;
    a.fp_off = 0;
    a.prefer_packed = p->prefer_packed;
    a.sparse = p->sparse;
    a.so_off = p->so_off;
    a.blob = p->blob;
    a.count = p->slot_ptr(slot);
    return a;
}

}  // namespace

extern "C" {

const char* smartgpu_version(void)
{
#ifdef SMARTGPU_AB
    return "smartgpu 0.2 (gfx950, A/B build)";
#else
    return "smartgpu 0.2 (gfx950)";
#endif
}
const char* smartgpu_last_error(void) { return g_error.c_str(); }

int smartgpu_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { set_error("hipGetDeviceCount failed"); return SMARTGPU_ERR_HIP; }
    return n;
}

int smartgpu_algo_id(const char* name)
{
    if (!name) return -1;
    std::string s(name);
    for (auto& ch : s) ch = static_cast<char>(std::tolower(static_cast<unsigned char>(ch)));
    for (int i = 0; i < SMARTGPU_NUM_ALGOS; ++i)
        if (s == kAlgoNames[i]) return i;
    return -1;
}

const char* smartgpu_algo_name(int algo)
{
    return (algo >= 0 && algo < SMARTGPU_NUM_ALGOS) ? kAlgoNames[algo] : nullptr;
}

int smartgpu_device_sync(int device)
{
    DeviceCtx* d = device_ctx(device);
    if (!d) return SMARTGPU_ERR_HIP;
    HIP_TRY(hipStreamSynchronize(d->stream), return SMARTGPU_ERR_HIP);
    return SMARTGPU_OK;
}

/* ---- text ------------------------------------------------------------- */
static smartgpu_text* text_upload_impl(const void* host, uint64_t n, int device, bool alphabet)
{
    if (!host && n) { set_error("host pointer is NULL"); return nullptr; }
    DeviceCtx* d = nullptr;
    smartgpu_text* t = text_alloc(n, device, &d);
    if (!t) return nullptr;
    const uint8_t* src = static_cast<const uint8_t*>(host);
    uint8_t* dst = t->base + sg::kFrontPad;
    for (uint64_t done = 0; done < n;) {  // pinned staging -> HBM
        const size_t chunk = static_cast<size_t>(std::min<uint64_t>(d->pinned_bytes, n - done));
        std::memcpy(d->pinned, src + done, chunk);
        if (hipMemcpyAsync(dst + done, d->pinned, chunk, hipMemcpyHostToDevice, d->stream) != hipSuccess ||
            hipStreamSynchronize(d->stream) != hipSuccess) {
            set_error("text upload failed at byte %llu", (unsigned long long)done);
            smartgpu_text_free(t);
            return nullptr;
        }
        done += chunk;
    }
    hipStreamSynchronize(d->stream);
    if (!(alphabet ? text_alphabet(t, d) : text_no_alphabet(t, d))) { smartgpu_text_free(t); return nullptr; }
    return t;
}

smartgpu_text* smartgpu_text_upload(const void* host, uint64_t n, int device) { return text_upload_impl(host, n, device, true); }

smartgpu_text* smartgpu_text_upload_tiled(const void* unit, uint64_t unit_len, uint64_t phase,
                                          uint64_t n, int device)
{
    if (!unit || unit_len == 0) { set_error("empty unit"); return nullptr; }
    smartgpu_text* u = smartgpu_text_upload(unit, unit_len, device);
    if (!u) return nullptr;
    DeviceCtx* d = nullptr;
    smartgpu_text* t = text_alloc(n, device, &d);
    if (!t) { smartgpu_text_free(u); return nullptr; }
    hipError_t e = sg::launch_tile_fill(t->base + sg::kFrontPad, u->data(), unit_len, phase % unit_len, n, d->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(d->stream);
    smartgpu_text_free(u);
    if (e != hipSuccess) {
        set_error("tile_fill failed: %s", hipGetErrorString(e));
        smartgpu_text_free(t);
        return nullptr;
    }
    if (!text_alphabet(t, d)) { smartgpu_text_free(t); return nullptr; }
    return t;
}

smartgpu_text* smartgpu_text_generate(uint64_t seed, int sigma, uint64_t off, uint64_t n, int device)
{
    if (sigma < 2 || sigma > 256) { set_error("sigma %d outside [2,256]", sigma); return nullptr; }
    DeviceCtx* d = nullptr;
    smartgpu_text* t = text_alloc(n, device, &d);
    if (!t) return nullptr;
    hipError_t e = sg::launch_generate(t->base + sg::kFrontPad, seed, sigma, off, n, d->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(d->stream);
    if (e != hipSuccess) {
        set_error("generate_text failed: %s", hipGetErrorString(e));
        smartgpu_text_free(t);
        return nullptr;
    }
    if (!text_alphabet(t, d)) { smartgpu_text_free(t); return nullptr; }
    return t;
}

void smartgpu_text_free(smartgpu_text* t)
{
    if (!t) return;
    if (t->base) {
        hipSetDevice(t->device);
        hipFree(t->base);
    }
    delete t;
}

uint64_t smartgpu_text_length(const smartgpu_text* t) { return t ? t->n : 0; }
int smartgpu_text_device(const smartgpu_text* t) { return t ? t->device : -1; }

int smartgpu_text_alphabet(const smartgpu_text* t, uint32_t bits[8])
{
    if (!t || !bits) { set_error("bad alphabet arguments"); return SMARTGPU_ERR_ARG; }
    std::memcpy(bits, t->alphabet, 32);
    return SMARTGPU_OK;
}

int smartgpu_text_read(const smartgpu_text* t, uint64_t off, uint64_t len, void* host)
{
    if (!t || !host || off > t->n || len > t->n - off) { set_error("bad text_read range"); return SMARTGPU_ERR_ARG; }
    DeviceCtx* d = device_ctx(t->device);
    if (!d) return SMARTGPU_ERR_HIP;
    HIP_TRY(hipStreamSynchronize(d->stream), return SMARTGPU_ERR_HIP);
    HIP_TRY(hipMemcpy(host, t->data() + off, len, hipMemcpyDeviceToHost), return SMARTGPU_ERR_HIP);
    return SMARTGPU_OK;
}

/* ---- plans ------------------------------------------------------------ */
smartgpu_plan* smartgpu_plan_create(int algo, const uint8_t* P, uint32_t m, int device)
{
    if (algo < 0 || algo >= SMARTGPU_NUM_ALGOS) { set_error("unknown algorithm id %d", algo); return nullptr; }
    if (!P || m < 1 || m > SMARTGPU_XSIZE) { set_error("pattern length %u outside [1,%d]", m, SMARTGPU_XSIZE); return nullptr; }
    if (m < min_pattern(algo)) { set_error("%s: not applicable for m < %u", kAlgoNames[algo], min_pattern(algo)); return nullptr; }
    DeviceCtx* d = device_ctx(device);
    if (!d) return nullptr;
    const double t0 = now_ms();
    smartgpu_plan* p = new smartgpu_plan;
    p->device = device;
    p->algo = algo;
    p->m = m;
    const std::vector<uint8_t> blob = build_blob(algo, P, m, &p->halo, &p->prefer_packed, &p->sparse, &p->so_off);
    bool ok = hipMalloc(reinterpret_cast<void**>(&p->blob), blob.size()) == hipSuccess &&
              hipMalloc(reinterpret_cast<void**>(&p->results), sizeof(unsigned long long) * sg::kResultSlots) == hipSuccess;
    if (ok) {
        std::memcpy(d->pinned, blob.data(), blob.size());
        ok = hipMemcpyAsync(p->blob, d->pinned, blob.size(), hipMemcpyHostToDevice, d->stream) == hipSuccess &&
             hipMemsetAsync(p->results, 0, sizeof(unsigned long long) * sg::kResultSlots, d->stream) == hipSuccess &&
             hipStreamSynchronize(d->stream) == hipSuccess;
    }
    if (!ok) {
        set_error("plan_create: device allocation or table upload failed (%s)", hipGetErrorString(hipGetLastError()));
        smartgpu_plan_free(p);
        return nullptr;
    }
    p->pre_ms = now_ms() - t0;
    return p;
}

void smartgpu_plan_free(smartgpu_plan* p)
{
    if (!p) return;
    hipSetDevice(p->device);
    for (int i = 0; i < sg::kResultSlots; ++i) {
        if (p->ev0[i]) hipEventDestroy(p->ev0[i]);
        if (p->ev1[i]) hipEventDestroy(p->ev1[i]);
    }
    if (p->blob) hipFree(p->blob);
    if (p->results) hipFree(p->results);
    delete p;
}

int smartgpu_plan_launch(smartgpu_plan* p, const smartgpu_text* text, uint64_t off, uint64_t n,
                         int slot, int timed)
{
    if (!p || !text) { set_error("plan or text is NULL"); return SMARTGPU_ERR_ARG; }
    if (slot < 0 || slot >= p->num_slots()) { set_error("slot %d out of range", slot); return SMARTGPU_ERR_ARG; }
    if (text->device != p->device) { set_error("plan is on device %d, text on %d", p->device, text->device); return SMARTGPU_ERR_ARG; }
    if (off > text->n || n > text->n - off) { set_error("range outside the text"); return SMARTGPU_ERR_ARG; }
    DeviceCtx* d = device_ctx(p->device);
    if (!d) return SMARTGPU_ERR_HIP;
    p->timed[slot] = timed != 0;
    if (timed) {
        if (!p->ev0[slot]) {
            HIP_TRY(hipEventCreate(&p->ev0[slot]), return SMARTGPU_ERR_HIP);
            HIP_TRY(hipEventCreate(&p->ev1[slot]), return SMARTGPU_ERR_HIP);
        }
        HIP_TRY(hipEventRecord(p->ev0[slot], d->stream), return SMARTGPU_ERR_HIP);
    }
    const sg::ScanArgs a = make_args(p, text, off, n, slot);
    HIP_TRY(sg::launch_scan(p->algo, a, d->num_cus, d->stream, text->codes()), return SMARTGPU_ERR_HIP);
    if (timed) HIP_TRY(hipEventRecord(p->ev1[slot], d->stream), return SMARTGPU_ERR_HIP);
    return SMARTGPU_OK;
}

int smartgpu_plan_result(smartgpu_plan* p, int slot, uint64_t* count, double* kernel_ms)
{
    if (!p || slot < 0 || slot >= p->num_slots()) { set_error("bad plan/slot"); return SMARTGPU_ERR_ARG; }
    DeviceCtx* d = device_ctx(p->device);
    if (!d) return SMARTGPU_ERR_HIP;
    HIP_TRY(hipMemcpyAsync(d->pinned_count, p->slot_ptr(slot), sizeof(unsigned long long), hipMemcpyDeviceToHost, d->stream),
            return SMARTGPU_ERR_HIP);
    HIP_TRY(hipStreamSynchronize(d->stream), return SMARTGPU_ERR_HIP);
    if (count_poisoned(*d->pinned_count)) return SMARTGPU_ERR_HIP;
    if (count) *count = *d->pinned_count;
    if (kernel_ms) {
        *kernel_ms = -1.0;
        if (p->timed[slot]) {
            float ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, p->ev0[slot], p->ev1[slot]), return SMARTGPU_ERR_HIP);
            *kernel_ms = ms;
        }
    }
    return SMARTGPU_OK;
}

const char* smartgpu_plan_kernel_name(const smartgpu_plan* p)
{
    return p ? sg::scan_kernel_name(p->algo, p->m, p->prefer_packed != 0, p->so_off != 0, p->halo) : nullptr;
}

const char* smartgpu_kernel_for(int algo, const uint8_t* P, uint32_t m)
{
    if (algo < 0 || algo >= SMARTGPU_NUM_ALGOS || !P || m < 1 || m > SMARTGPU_XSIZE || m < min_pattern(algo)) {
        set_error("kernel_for: algorithm %d / pattern length %u not applicable", algo, m);
        return nullptr;
    }
    uint32_t halo = 0, prefer_packed = 0, sparse = 0, so_off = 0;
    (void)build_blob(algo, P, m, &halo, &prefer_packed, &sparse, &so_off);
    return sg::scan_kernel_name(algo, m, prefer_packed != 0, so_off != 0, halo);
}

void* smartgpu_plan_result_device_ptr(smartgpu_plan* p) { return p ? p->slot_ptr(0) : nullptr; }

int smartgpu_plan_reset(smartgpu_plan* p)
{
    if (!p) { set_error("plan is NULL"); return SMARTGPU_ERR_ARG; }
    DeviceCtx* d = device_ctx(p->device);
    if (!d) return SMARTGPU_ERR_HIP;
    HIP_TRY(hipMemsetAsync(p->slot_ptr(0), 0, sizeof(unsigned long long) * p->num_slots(), d->stream),
            return SMARTGPU_ERR_HIP);
    return SMARTGPU_OK;
}

int smartgpu_plan_set_result_buffer(smartgpu_plan* p, void* device_u64, int nslots)
{
    if (!p || (device_u64 && (nslots < 1 || nslots > sg::kResultSlots))) { set_error("bad result buffer"); return SMARTGPU_ERR_ARG; }
    p->ext_results = static_cast<unsigned long long*>(device_u64);
    p->ext_slots = device_u64 ? nslots : 0;
    return SMARTGPU_OK;
}

int smartgpu_stream_mark(int device, int which)
{
    DeviceCtx* d = device_ctx(device);
    if (!d || which < 0 || which > 1) return SMARTGPU_ERR_ARG;
    if (!d->mark[which]) HIP_TRY(hipEventCreate(&d->mark[which]), return SMARTGPU_ERR_HIP);
    HIP_TRY(hipEventRecord(d->mark[which], d->stream), return SMARTGPU_ERR_HIP);
    return SMARTGPU_OK;
}

int smartgpu_stream_elapsed_ms(int device, double* ms)
{
    DeviceCtx* d = device_ctx(device);
    if (!d || !ms || !d->mark[0] || !d->mark[1]) { set_error("stream marks not set"); return SMARTGPU_ERR_ARG; }
    HIP_TRY(hipEventSynchronize(d->mark[1]), return SMARTGPU_ERR_HIP);
    float f = 0.f;
    HIP_TRY(hipEventElapsedTime(&f, d->mark[0], d->mark[1]), return SMARTGPU_ERR_HIP);
    *ms = f;
    return SMARTGPU_OK;
}

int smartgpu_probe_read_ms(const smartgpu_text* t, int reps, double* ms_per_pass)
{
    if (!t || reps < 1 || !ms_per_pass) { set_error("bad probe arguments"); return SMARTGPU_ERR_ARG; }
    DeviceCtx* d = device_ctx(t->device);
    if (!d) return SMARTGPU_ERR_HIP;
    unsigned long long* sink = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&sink), 8), return SMARTGPU_ERR_HIP);
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0), return SMARTGPU_ERR_HIP);
    HIP_TRY(hipEventCreate(&e1), return SMARTGPU_ERR_HIP);
    sg::launch_probe_read(t->data(), t->n, sink, d->num_cus, d->stream);  // warm-up
    hipEventRecord(e0, d->stream);
    for (int i = 0; i < reps; ++i) sg::launch_probe_read(t->data(), t->n, sink, d->num_cus, d->stream);
    hipEventRecord(e1, d->stream);
    HIP_TRY(hipEventSynchronize(e1), return SMARTGPU_ERR_HIP);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    *ms_per_pass = ms / reps;
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    hipFree(sink);
    return SMARTGPU_OK;
}

int smartgpu_tune(int key, int value)
{
    if (key < 0 || key >= 8) { set_error("tune key %d out of range", key); return SMARTGPU_ERR_ARG; }
    if (!sg::tune_supported(key, value)) {
        set_error("tune(%d,%d) selects a superseded kernel: only in the A/B build (make -C smart_amd/csrc AB=1 -> libsmartgpu_ab.so)", key, value);
        return SMARTGPU_ERR_ARG;
    }
    sg::g_tune[key] = value;
    return SMARTGPU_OK;
}

void* smartgpu_stream_handle(int device)
{
    DeviceCtx* d = device_ctx(device);
    return d ? static_cast<void*>(d->stream) : nullptr;
}

/* ---- one-shot searches ------------------------------------------------- */
int smartgpu_search64(int algo, const uint8_t* P, uint32_t m, const smartgpu_text* text,
                      uint64_t off, uint64_t n, uint64_t* count, double* pre_ms, double* run_ms)
{
    const int rc = check_search_args(algo, P, m, text, off, n);
    if (rc != SMARTGPU_OK) return rc;
    // a pattern set of one: tables through the device's arena and staging buffer (no allocation per call — a plan's
    // two hipMalloc and its memset were most of the 25-30 us a call cost around a 3 us kernel), one launch, one read-back
    const uint8_t* set[1] = {P};
    uint64_t c = 0;
    double pre = 0.0, wall = 0.0;
    const int r = smartgpu_search_batch64(algo, set, m, 1, text, off, n, &c, &pre, nullptr, &wall);
    if (r != SMARTGPU_OK) return r;
    g_last_pre_ms = pre;
    g_last_run_ms = wall;  // launch -> count on host, the run_time analogue (main.h:29,31)
    if (count) *count = c;
    if (pre_ms) *pre_ms = pre;
    if (run_ms) *run_ms = wall;
    return SMARTGPU_OK;
}


/* ---- a whole pattern set per call (the harness loop, src/smart.c:312-345) ------------- */
}  // extern "C"

namespace {

// make sure the device's batch arena holds `blob_bytes` of tables and `k` counts
bool batch_reserve(DeviceCtx* d, size_t blob_bytes, size_t k)
{
    if (blob_bytes > d->arena_bytes) {
        if (d->arena) (void)hipFree(d->arena);
        d->arena = nullptr;
        d->arena_bytes = 0;
        const size_t want = std::max(blob_bytes + blob_bytes / 4, size_t(8) << 20);
        if (hipMalloc(reinterpret_cast<void**>(&d->arena), want) != hipSuccess) { set_error("batch arena: hipMalloc of %zu bytes failed", want); return false; }
        d->arena_bytes = want;
    }
    if (k > d->batch_slots) {
        if (d->batch_counts) (void)hipFree(d->batch_counts);
        if (d->pinned_counts) (void)hipHostFree(d->pinned_counts);
        d->batch_counts = nullptr;
        d->pinned_counts = nullptr;
        d->batch_slots = 0;
        const size_t want = std::max<size_t>(k, 1024);
        if (hipMalloc(reinterpret_cast<void**>(&d->batch_counts), want * 8) != hipSuccess ||
            hipHostMalloc(reinterpret_cast<void**>(&d->pinned_counts), want * 8, hipHostMallocDefault) != hipSuccess) {
            set_error("batch counts: allocation of %zu slots failed", want);
            return false;
        }
        d->batch_slots = want;
    }
    return true;
}

struct BatchPlan { uint32_t halo, prefer_packed, sparse, so_off; size_t off; };
// the staging buffer of a set of K patterns: [table blobs ...][K argument records][K launch-order indices]
size_t batch_tail_offset(const DeviceCtx* d, uint32_t K)
{
    return (d->pinned_bytes - static_cast<size_t>(K) * (4 + sizeof(sg::BatchItem))) & ~size_t(63);
}
constexpr uint64_t kOneGridMaxText = 32ull << 20;  // pattern sets over texts up to this size run as ONE grid per kernel
constexpr uint32_t kOneGridMaxY = 65535;            // HIP's limit on gridDim.y: patterns per grid of the one-grid form
constexpr uint32_t kBatchMaxPatterns = 1u << 18;    // per call (argument records and launch order of a set sit in the staging buffer's tail: 9 of its 32 MB)

// Build the K blobs on the host ONCE and place them in the arena of every device of `ds` (one device: a plain pattern
// set; several: the shards of a multi-GPU text — VERDICT r3: the tables of a set were rebuilt per device, 8x the
// preprocessing at 8 GPUs).  pre_ms[k] = host table construction of pattern k + its share of the uploads.
// plans[k].off = offset of blob k in the arenas (the same in each).  The blobs are staged in ds[0]'s pinned buffer and
// copied from there to every arena, each copy on its device's own stream.
int batch_upload(const std::vector<DeviceCtx*>& ds, int algo, const uint8_t* const* P, uint32_t m, uint32_t K, std::vector<BatchPlan>& plans,
                 double* pre_ms)
{
    static thread_local std::vector<uint8_t> blob;  // one buffer for every pattern of every set
    DeviceCtx* const d0 = ds[0];
    std::vector<double> host_ms(K, 0.0);
    plans.resize(K);
    // the blobs of one algorithm and one length are equally long (multiples of 256): size the arena from the first
    if (!P[0]) { set_error("pattern 0 is NULL"); return SMARTGPU_ERR_ARG; }
    build_blob(blob, algo, P[0], m, &plans[0].halo, &plans[0].prefer_packed, &plans[0].sparse, &plans[0].so_off);
    const size_t room = batch_tail_offset(d0, K);  // the tail of the staging buffer holds the set's argument records and launch order
    if (blob.size() > room) { set_error("a table blob of %zu bytes exceeds the staging buffer", blob.size()); return SMARTGPU_ERR_NOMEM; }
    size_t arena_min = 0;
    for (DeviceCtx* d : ds) {
        if (d->device != d0->device) HIP_TRY(hipSetDevice(d->device), return SMARTGPU_ERR_HIP);
        if (!batch_reserve(d, (blob.size() + 4096) * K + 256 + K * sizeof(sg::BatchItem), K)) return SMARTGPU_ERR_NOMEM;
        arena_min = arena_min ? std::min(arena_min, d->arena_bytes) : d->arena_bytes;
    }
    double up_total = 0.0;
    size_t total = 0, fill = 0, fill_off = 0;  // staging holds `fill` bytes that belong at arena offset fill_off
    // the copy is waited for only where the staging buffer is filled again in this call: the launches that follow are
    // ordered behind it on each device's stream, and every caller ends with a synchronisation before the buffer is reused
    auto flush = [&](bool more) -> bool {
        if (!fill) return true;
        const double t_up = now_ms();
        bool ok = true;
        for (DeviceCtx* d : ds) {
            if (ds.size() > 1) ok = ok && hipSetDevice(d->device) == hipSuccess;
            ok = ok && hipMemcpyAsync(d->arena + fill_off, d0->pinned, fill, hipMemcpyHostToDevice, d->stream) == hipSuccess;
        }
        if (more || ds.size() > 1)  // several devices read ONE staging buffer: all done before the caller's own use of it
            for (DeviceCtx* d : ds) ok = ok && hipStreamSynchronize(d->stream) == hipSuccess;
        up_total += now_ms() - t_up;
        fill_off += fill;
        fill = 0;
        return ok;
    };
    for (uint32_t k = 0; k < K; ++k) {
        if (!P[k]) { set_error("pattern %u is NULL", k); return SMARTGPU_ERR_ARG; }
        const double t0 = now_ms();
        if (k) build_blob(blob, algo, P[k], m, &plans[k].halo, &plans[k].prefer_packed, &plans[k].sparse, &plans[k].so_off);
        if (total + blob.size() + 256 + K * sizeof(sg::BatchItem) > arena_min) {  // rerouted patterns carry masks the first did not
            set_error("batch: the table arena (%zu bytes) is too small for this pattern set", arena_min);
            return SMARTGPU_ERR_NOMEM;
        }
        if (fill + blob.size() > room && !flush(true)) { set_error("batch: table upload failed (%s)", hipGetErrorString(hipGetLastError())); return SMARTGPU_ERR_HIP; }
        std::memcpy(d0->pinned + fill, blob.data(), blob.size());  // straight into the staging buffer
        fill += blob.size();
        plans[k].off = total;
        total += blob.size();  // multiples of 256
        host_ms[k] = now_ms() - t0;
    }
    if (!flush(false)) { set_error("batch: table upload failed (%s)", hipGetErrorString(hipGetLastError())); return SMARTGPU_ERR_HIP; }
    if (ds.size() > 1) HIP_TRY(hipSetDevice(d0->device), return SMARTGPU_ERR_HIP);
    const double up_ms = up_total / K;
    if (pre_ms)
        for (uint32_t k = 0; k < K; ++k) pre_ms[k] = host_ms[k] + up_ms;
    return SMARTGPU_OK;
}

sg::ScanArgs batch_args(const BatchPlan& bp, const DeviceCtx* d, uint32_t m, const smartgpu_text* text, uint64_t off, uint64_t n,
                        unsigned long long* slot)
{
    sg::ScanArgs a;
    a.text = text->data();
    a.s_begin = off;
    a.s_end = (n >= m) ? off + n - m + 1 : off;
    a.m = m;
    a.halo = bp.halo;
    a.fp_off = 0;
    a.prefer_packed = bp.prefer_packed;
    a.sparse = bp.sparse;
    a.so_off = bp.so_off;
    a.blob = d->arena + bp.off;
    a.count = slot;
    return a;
}


// Enqueue the searches of a pattern set on the device's stream; counts go to d->batch_counts[0..K).
// Small texts (SMART's stock 1 MiB: a search is 64 workgroups and 3 us, less than its launch costs the host):
// the set runs as ONE grid per kernel, gridDim.y = patterns (sg::launch_scan_set) — the per-pattern arguments
// go up as one array behind the tables; patterns are grouped by the kernel and grid their plans choose.
// Larger texts: K launches back to back (each fills the chip on its own), with one HIP event per pattern
// if per-pattern device times are wanted.  run_ms (or NULL): device time per pattern; in the one-grid form
// a group's time divided by its patterns.
int batch_enqueue(DeviceCtx* d, int algo, const std::vector<BatchPlan>& plans, uint32_t m, const smartgpu_text* text,
                  uint64_t off, uint64_t n, uint32_t K, bool timed, std::vector<std::pair<uint32_t, uint32_t>>* groups_out,
                  bool each = false)  // each: one launch per pattern whatever the text's size (smartgpu_search_batch64_each)
{
    HIP_TRY(hipMemsetAsync(d->batch_counts, 0, static_cast<size_t>(K) * 8, d->stream), return SMARTGPU_ERR_HIP);
    if (timed)
        while (d->batch_events.size() < static_cast<size_t>(K) + 1) {
            hipEvent_t e;
            HIP_TRY(hipEventCreate(&e), return SMARTGPU_ERR_HIP);
            d->batch_events.push_back(e);
        }
    if (groups_out) groups_out->clear();
    if (n > kOneGridMaxText || K == 1 || each) {
        if (timed) HIP_TRY(hipEventRecord(d->batch_events[0], d->stream), return SMARTGPU_ERR_HIP);
        for (uint32_t k = 0; k < K; ++k) {
            const sg::ScanArgs a = batch_args(plans[k], d, m, text, off, n, d->batch_counts + k);
            HIP_TRY(sg::launch_scan(algo, a, d->num_cus, d->stream, text->codes()), return SMARTGPU_ERR_HIP);
            if (timed) HIP_TRY(hipEventRecord(d->batch_events[k + 1], d->stream), return SMARTGPU_ERR_HIP);
            if (groups_out) groups_out->push_back({k, 1u});
        }
        return SMARTGPU_OK;
    }
    // one grid per group of patterns whose plans lead to the same kernel and grid
    // (the key is what launch_scan derives kernel, template arguments and grid from: prefer_packed — for KMP the window
    // of its table —, the Shift-Or reroute, sparse, and for BNDM / BNDML the q of bndm_scan that travels as halo;
    // a mixed set ran every BNDM pattern with the first pattern's q)
    // (... and for Horspool the q of its q-gram table, bits 8.. of halo)
    const bool halo_is_q = algo == SMARTGPU_BNDM || algo == SMARTGPU_BNDML;
    auto key = [&](uint32_t k) {
        const uint32_t pp = algo == SMARTGPU_KMP ? plans[k].prefer_packed : (plans[k].prefer_packed ? 1u : 0u);
        const uint32_t variant = halo_is_q ? plans[k].halo & 0x1FFu : algo == SMARTGPU_HOR ? (plans[k].halo >> 8) & 0xFFu : 0u;  // BNDM: q and the gram-window mark
        return (static_cast<uint64_t>(pp) << 20) | (variant << 8) | (plans[k].so_off ? 2u : 0u) | (plans[k].sparse ? 1u : 0u);
    };
    std::vector<uint32_t> order(K);
    for (uint32_t k = 0; k < K; ++k) order[k] = k;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return key(x) < key(y); });
    // (in the staging buffer's tail: its front may still be on its way to the arena — batch_upload does not wait)
    sg::BatchItem* host_items = reinterpret_cast<sg::BatchItem*>(d->pinned + batch_tail_offset(d, K));
    for (uint32_t i = 0; i < K; ++i) {
        const BatchPlan& bp = plans[order[i]];
        const sg::ScanArgs pa = sg::prepare_scan_args(algo, batch_args(bp, d, m, text, off, n, d->batch_counts));
        host_items[i] = sg::BatchItem{bp.off, order[i], pa.halo, pa.fp_off, pa.prefer_packed, pa.sparse, pa.so_off};
    }
    sg::BatchItem* dev_items = reinterpret_cast<sg::BatchItem*>(d->arena + d->arena_bytes - ((static_cast<size_t>(K) * sizeof(sg::BatchItem) + 255) & ~size_t(255)));
    HIP_TRY(hipMemcpyAsync(dev_items, host_items, static_cast<size_t>(K) * sizeof(sg::BatchItem), hipMemcpyHostToDevice, d->stream),
            return SMARTGPU_ERR_HIP);
    uint32_t ev = 0;
    if (timed) HIP_TRY(hipEventRecord(d->batch_events[ev++], d->stream), return SMARTGPU_ERR_HIP);
    for (uint32_t i = 0; i < K;) {
        uint32_t j = i;
        while (j < K && key(order[j]) == key(order[i])) ++j;
        // the set's common arguments: blob = arena base, count = first slot (the items add their own), and the
        // group's plan fields, which choose kernel and grid
        sg::ScanArgs first = batch_args(plans[order[i]], d, m, text, off, n, d->batch_counts);
        first.blob = d->arena;
        // gridDim.y holds at most 65535: a larger group goes as several grids over slices of its items
        for (uint32_t lo = i; lo < j; lo += kOneGridMaxY)
            HIP_TRY(sg::launch_scan_set(algo, first, dev_items + lo, std::min(kOneGridMaxY, j - lo), d->num_cus, d->stream, text->codes()), return SMARTGPU_ERR_HIP);
        if (timed) HIP_TRY(hipEventRecord(d->batch_events[ev++], d->stream), return SMARTGPU_ERR_HIP);
        if (groups_out) groups_out->push_back({i, j - i});
        i = j;
    }
    // groups_out indexes the sorted order; remember it for the caller through the pinned array's tail
    if (groups_out) {
        uint32_t* ord = reinterpret_cast<uint32_t*>(d->pinned + d->pinned_bytes - static_cast<size_t>(K) * 4);
        for (uint32_t i = 0; i < K; ++i) ord[i] = order[i];
    }
    return SMARTGPU_OK;
}

}  // namespace

extern "C" {

static int search_batch_impl(int algo, const uint8_t* const* P, uint32_t m, uint32_t K, const smartgpu_text* text,
                             uint64_t off, uint64_t n, uint64_t* counts, double* pre_ms, double* run_ms, double* batch_ms, bool each)
{
    if (!P || K < 1 || !counts) { set_error("batch: P/counts NULL or K = 0"); return SMARTGPU_ERR_ARG; }
    if (K > kBatchMaxPatterns) { set_error("batch: %u patterns in one set (at most %u)", K, kBatchMaxPatterns); return SMARTGPU_ERR_ARG; }
    const int rc = check_search_args(algo, P[0], m, text, off, n);
    if (rc != SMARTGPU_OK) return rc;
    DeviceCtx* d = device_ctx(text->device);
    if (!d) return SMARTGPU_ERR_HIP;
    std::vector<BatchPlan> plans;
    const int up = batch_upload({d}, algo, P, m, K, plans, pre_ms);  // preprocessing phase
    if (up != SMARTGPU_OK) return up;
    const bool timed = run_ms != nullptr;
    // searching phase: the set's launches, one read-back
    std::vector<std::pair<uint32_t, uint32_t>> groups;
    const double t0 = now_ms();
    const int eq = batch_enqueue(d, algo, plans, m, text, off, n, K, timed, &groups, each);
    if (eq != SMARTGPU_OK) return eq;
    HIP_TRY(hipMemcpyAsync(d->pinned_counts, d->batch_counts, static_cast<size_t>(K) * 8, hipMemcpyDeviceToHost, d->stream),
            return SMARTGPU_ERR_HIP);
    HIP_TRY(hipStreamSynchronize(d->stream), return SMARTGPU_ERR_HIP);
    const double wall = now_ms() - t0;
    for (uint32_t k = 0; k < K; ++k) {
        if (count_poisoned(d->pinned_counts[k])) return SMARTGPU_ERR_HIP;
        counts[k] = d->pinned_counts[k];
    }
    if (timed) {
        const bool one_grid = !(n > kOneGridMaxText || K == 1 || each);
        const uint32_t* ord = reinterpret_cast<const uint32_t*>(d->pinned + d->pinned_bytes - static_cast<size_t>(K) * 4);
        for (size_t g = 0; g < groups.size(); ++g) {
            float ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, d->batch_events[g], d->batch_events[g + 1]), return SMARTGPU_ERR_HIP);
            for (uint32_t i = groups[g].first; i < groups[g].first + groups[g].second; ++i)
                run_ms[one_grid ? ord[i] : i] = ms / groups[g].second;
        }
    }
    if (batch_ms) *batch_ms = wall;
    g_last_pre_ms = pre_ms ? pre_ms[K - 1] : 0.0;
    g_last_run_ms = wall / K;
    return SMARTGPU_OK;
}

int smartgpu_search_batch64(int algo, const uint8_t* const* P, uint32_t m, uint32_t K, const smartgpu_text* text,
                            uint64_t off, uint64_t n, uint64_t* counts, double* pre_ms, double* run_ms, double* batch_ms)
{
    return search_batch_impl(algo, P, m, K, text, off, n, counts, pre_ms, run_ms, batch_ms, false);
}

int smartgpu_search_batch64_each(int algo, const uint8_t* const* P, uint32_t m, uint32_t K, const smartgpu_text* text,
                                 uint64_t off, uint64_t n, uint64_t* counts, double* pre_ms, double* run_ms, double* batch_ms)
{
    return search_batch_impl(algo, P, m, K, text, off, n, counts, pre_ms, run_ms, batch_ms, true);
}

/* ---- occurrence positions ------------------------------------------------- */
int smartgpu_find64(const uint8_t* P, uint32_t m, const smartgpu_text* text, uint64_t off, uint64_t n,
                    uint64_t* positions, uint64_t cap, uint64_t* count)
{
    const int rc = check_search_args(SMARTGPU_EPSM, P, m, text, off, n);
    if (rc != SMARTGPU_OK) return rc;
    if (!count || (cap && !positions)) { set_error("find64: count/positions must not be NULL"); return SMARTGPU_ERR_ARG; }
    DeviceCtx* d = device_ctx(text->device);
    if (!d) return SMARTGPU_ERR_HIP;
    smartgpu_plan* p = smartgpu_plan_create(SMARTGPU_EPSM, P, m, text->device);
    if (!p) return SMARTGPU_ERR_HIP;
    unsigned long long* out = nullptr;
    int r = SMARTGPU_OK;
    if (cap && hipMalloc(reinterpret_cast<void**>(&out), cap * sizeof(unsigned long long)) != hipSuccess) {
        set_error("find64: cannot allocate %llu positions on the device", (unsigned long long)cap);
        r = SMARTGPU_ERR_NOMEM;
    }
    unsigned long long total = 0;
    if (r == SMARTGPU_OK) {
        sg::ScanArgs a = make_args(p, text, off, n, 0);
        a.fp_off = sg::kTableOff;  // EPSM blob: the fingerprint follows the pattern slot
        bool ok = hipMemsetAsync(a.count, 0, sizeof(unsigned long long), d->stream) == hipSuccess &&
                  sg::launch_find(a, out, cap, d->num_cus, d->stream) == hipSuccess &&
                  hipMemcpyAsync(d->pinned_count, a.count, sizeof(unsigned long long), hipMemcpyDeviceToHost, d->stream) == hipSuccess &&
                  hipStreamSynchronize(d->stream) == hipSuccess;
        if (ok) {
            total = *d->pinned_count;
            const unsigned long long have = total < cap ? total : cap;
            if (have) ok = hipMemcpy(positions, out, have * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess;
            if (ok && have) std::sort(positions, positions + have);  // waves append in no particular order
        }
        if (!ok) {
            set_error("find64: %s", hipGetErrorString(hipGetLastError()));
            r = SMARTGPU_ERR_HIP;
        }
    }
    if (out) (void)hipFree(out);
    smartgpu_plan_free(p);
    if (r != SMARTGPU_OK) return r;
    *count = total;
    if (total > cap) { set_error("find64: %llu occurrences, room for %llu", total, (unsigned long long)cap); return SMARTGPU_ERR_NOMEM; }
    return SMARTGPU_OK;
}

void smartgpu_last_times(double* pre_ms, double* run_ms)
{
    if (pre_ms) *pre_ms = g_last_pre_ms;
    if (run_ms) *run_ms = g_last_run_ms;
}

static int search_host(int algo, const unsigned char* P, int m, const unsigned char* T, int n)
{
    if (!P || !T || m < 1 || n < 0) return SMARTGPU_NA;
    if (m > SMARTGPU_XSIZE) return SMARTGPU_NA;
    smartgpu_text* t = text_upload_impl(T, static_cast<uint64_t>(n), 0, false);  // searched once: no alphabet pass
    if (!t) return SMARTGPU_NA;
    uint64_t c = 0;
    const int rc = smartgpu_search64(algo, P, static_cast<uint32_t>(m), t, 0, static_cast<uint64_t>(n), &c, nullptr, nullptr);
    smartgpu_text_free(t);
    if (rc != SMARTGPU_OK || c > 0x7FFFFFFFull) return SMARTGPU_NA;
    return static_cast<int>(c);
}

int smartgpu_hor_search(const unsigned char* P, int m, const unsigned char* T, int n) { return search_host(SMARTGPU_HOR, P, m, T, n); }
int smartgpu_bm_search(const unsigned char* P, int m, const unsigned char* T, int n) { return search_host(SMARTGPU_BM, P, m, T, n); }
int smartgpu_kmp_search(const unsigned char* P, int m, const unsigned char* T, int n) { return search_host(SMARTGPU_KMP, P, m, T, n); }
int smartgpu_so_search(const unsigned char* P, int m, const unsigned char* T, int n) { return search_host(SMARTGPU_SO, P, m, T, n); }
int smartgpu_bndm_search(const unsigned char* P, int m, const unsigned char* T, int n) { return search_host(SMARTGPU_BNDM, P, m, T, n); }
int smartgpu_epsm_search(const unsigned char* P, int m, const unsigned char* T, int n) { return search_host(SMARTGPU_EPSM, P, m, T, n); }
int smartgpu_sa_search(const unsigned char* P, int m, const unsigned char* T, int n) { return search_host(SMARTGPU_SA, P, m, T, n); }
int smartgpu_qs_search(const unsigned char* P, int m, const unsigned char* T, int n) { return search_host(SMARTGPU_QS, P, m, T, n); }
int smartgpu_tunedbm_search(const unsigned char* P, int m, const unsigned char* T, int n) { return search_host(SMARTGPU_TUNEDBM, P, m, T, n); }
int smartgpu_raita_search(const unsigned char* P, int m, const unsigned char* T, int n) { return search_host(SMARTGPU_RAITA, P, m, T, n); }
int smartgpu_hash3_search(const unsigned char* P, int m, const unsigned char* T, int n) { return search_host(SMARTGPU_HASH3, P, m, T, n); }
int smartgpu_hash5_search(const unsigned char* P, int m, const unsigned char* T, int n) { return search_host(SMARTGPU_HASH5, P, m, T, n); }
int smartgpu_hash8_search(const unsigned char* P, int m, const unsigned char* T, int n) { return search_host(SMARTGPU_HASH8, P, m, T, n); }
int smartgpu_sbndm_search(const unsigned char* P, int m, const unsigned char* T, int n) { return search_host(SMARTGPU_SBNDM, P, m, T, n); }
int smartgpu_kr_search(const unsigned char* P, int m, const unsigned char* T, int n) { return search_host(SMARTGPU_KR, P, m, T, n); }
int smartgpu_bndml_search(const unsigned char* P, int m, const unsigned char* T, int n) { return search_host(SMARTGPU_BNDML, P, m, T, n); }

/* ---- one process, several GPUs ------------------------------------------ */
}  // extern "C"

namespace {

// RCCL is loaded on first use (dlopen), so the library itself has no link-time
// dependency on it; only the multi-GPU reduce needs it.  Signatures, ncclUint64 and ncclSum
// come from <rccl/rccl.h>: an ABI change there is a compile error here, not a silent miscount.
struct Rccl {
    void* lib = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool load()
    {
        if (lib) return true;
        // (HSA_ENABLE_IPC_MODE_LEGACY=0 — dmabuf IPC, the only mode this pool's driver supports — is an HSA runtime
        // flag read at hsa_init: set by this library's constructor below, before any HIP call of the process can be ours)
        lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!lib) { set_error("cannot load librccl.so: %s", dlerror()); return false; }
        CommInitAll = reinterpret_cast<decltype(CommInitAll)>(dlsym(lib, "ncclCommInitAll"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
        AllReduce = reinterpret_cast<decltype(AllReduce)>(dlsym(lib, "ncclAllReduce"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(dlsym(lib, "ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(dlsym(lib, "ncclGroupEnd"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
        if (!CommInitAll || !CommDestroy || !AllReduce || !GroupStart || !GroupEnd || !GetErrorString) {
            set_error("librccl.so lacks an expected symbol");
            return false;
        }
        return true;
    }
};
Rccl g_rccl;

// Runs when the library is loaded, i.e. before its first HIP call: ROCr reads the flag when it initialises (hsa_init,
// on the process's first HIP call).  A process that initialised HIP before loading this library has to set it itself
// (bench.py and host/smart.c do so at entry); an explicit setting in the environment is never overridden.
__attribute__((constructor)) void smartgpu_set_ipc_mode() { setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0); }

}  // namespace

struct smartgpu_mtext {
    uint64_t n = 0;
    std::vector<int> devices;
    std::vector<smartgpu_text*> shards;   // shard g = bytes [begin[g], begin[g+1] + overlap)
    std::vector<uint64_t> begin;          // k+1 entries, start-position ownership
    std::vector<ncclComm_t> comms;        // RCCL communicators, created on first RCCL reduce
};

namespace {

// first byte shard g of k owns (the shards' sizes differ by at most one byte; g = k: n)
uint64_t shard_begin(uint64_t n, int k, int g) { return n / k * g + std::min<uint64_t>(g, n % k); }

smartgpu_mtext* mtext_new(uint64_t n, int ngpus, const int* devices)
{
    if (ngpus < 1 || ngpus > kMaxDevices) { set_error("ngpus %d outside [1,%d]", ngpus, kMaxDevices); return nullptr; }
    smartgpu_mtext* t = new smartgpu_mtext;
    t->n = n;
    for (int g = 0; g < ngpus; ++g) t->devices.push_back(devices ? devices[g] : g);
    for (int g = 0; g <= ngpus; ++g) t->begin.push_back(shard_begin(n, ngpus, g));
    return t;
}

// bytes shard g has to hold: its own starts plus XSIZE-1 bytes of the next shards
void shard_span(const smartgpu_mtext* t, int g, uint64_t* off, uint64_t* len)
{
    uint64_t own = 0;
    (void)smartgpu_mtext_partition(t->n, static_cast<int>(t->devices.size()), g, off, &own, len);
}

// k-1 parked host threads, one per further device of a multi-GPU search: the caller enqueues device 0's shard itself
// while worker g enqueues device g's (one host thread enqueued the k devices' launches one after the other —
// with one pattern on 128 MiB shards, a 28 us kernel, the launch latency of eight devices in a row was the bound).
// The threads are created on the first multi-GPU search and parked on a condition variable between calls.
class LaunchPool {
public:
    // fn(g) for g = 0 .. k-1: g = 0 on the calling thread, the others on the workers; returns when all are done
    void run(int k, const std::function<void(int)>& fn)
    {
        if (k <= 1) { if (k == 1) fn(0); return; }
        {
            std::unique_lock<std::mutex> lk(mu_);
            while (static_cast<int>(workers_.size()) < k - 1) {
                const int idx = static_cast<int>(workers_.size()) + 1;
                workers_.emplace_back([this, idx] { loop(idx); });
            }
            job_ = &fn;
            active_ = k;
            pending_ = k - 1;
            ++gen_;
        }
        go_.notify_all();
        fn(0);
        std::unique_lock<std::mutex> lk(mu_);
        done_.wait(lk, [this] { return pending_ == 0; });
        job_ = nullptr;
    }
    ~LaunchPool()
    {
        {
            std::unique_lock<std::mutex> lk(mu_);
            stop_ = true;
        }
        go_.notify_all();
        for (std::thread& t : workers_) t.join();
    }

private:
    void loop(int idx)
    {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void(int)>* job = nullptr;
            {
                std::unique_lock<std::mutex> lk(mu_);
                go_.wait(lk, [&] { return stop_ || gen_ != seen; });
                if (stop_) return;
                seen = gen_;
                if (idx < active_) job = job_;
            }
            if (!job) continue;  // this call uses fewer devices than there are workers
            (*job)(idx);
            std::unique_lock<std::mutex> lk(mu_);
            if (--pending_ == 0) done_.notify_one();
        }
    }
    std::mutex mu_;
    std::condition_variable go_, done_;
    std::vector<std::thread> workers_;
    const std::function<void(int)>* job_ = nullptr;
    uint64_t gen_ = 0;
    int active_ = 0, pending_ = 0;
    bool stop_ = false;
};
LaunchPool g_pool;

}  // namespace

extern "C" {

smartgpu_mtext* smartgpu_mtext_upload(const void* host, uint64_t n, int ngpus, const int* devices)
{
    if (!host && n) { set_error("host pointer is NULL"); return nullptr; }
    smartgpu_mtext* t = mtext_new(n, ngpus, devices);
    if (!t) return nullptr;
    for (int g = 0; g < ngpus; ++g) {
        uint64_t off, len;
        shard_span(t, g, &off, &len);
        smartgpu_text* s = smartgpu_text_upload(static_cast<const uint8_t*>(host) + off, len, t->devices[g]);
        if (!s) { smartgpu_mtext_free(t); return nullptr; }
        t->shards.push_back(s);
    }
    return t;
}

smartgpu_mtext* smartgpu_mtext_generate(uint64_t seed, int sigma, uint64_t n, int ngpus, const int* devices)
{
    smartgpu_mtext* t = mtext_new(n, ngpus, devices);
    if (!t) return nullptr;
    for (int g = 0; g < ngpus; ++g) {
        uint64_t off, len;
        shard_span(t, g, &off, &len);
        smartgpu_text* s = smartgpu_text_generate(seed, sigma, off, len, t->devices[g]);
        if (!s) { smartgpu_mtext_free(t); return nullptr; }
        t->shards.push_back(s);
    }
    return t;
}

void smartgpu_mtext_free(smartgpu_mtext* t)
{
    if (!t) return;
    for (smartgpu_text* s : t->shards) smartgpu_text_free(s);
    for (ncclComm_t c : t->comms)
        if (c && g_rccl.CommDestroy) g_rccl.CommDestroy(c);
    delete t;
}

uint64_t smartgpu_mtext_length(const smartgpu_mtext* t) { return t ? t->n : 0; }
int smartgpu_mtext_ngpus(const smartgpu_mtext* t) { return t ? static_cast<int>(t->devices.size()) : 0; }

int smartgpu_mtext_partition(uint64_t n, int ngpus, int g, uint64_t* begin, uint64_t* own, uint64_t* held)
{
    if (ngpus < 1 || ngpus > kMaxDevices || g < 0 || g >= ngpus) { set_error("partition: shard %d of %d", g, ngpus); return SMARTGPU_ERR_ARG; }
    const uint64_t b = shard_begin(n, ngpus, g), e = shard_begin(n, ngpus, g + 1);
    if (begin) *begin = b;
    if (own) *own = e - b;
    if (held) *held = std::min<uint64_t>(n, e + SMARTGPU_XSIZE - 1) - b;
    return SMARTGPU_OK;
}

int smartgpu_selftest_launch_pool(int k, int rounds)
{
    // every round: fn(g) adds g + 1 into a slot of its own and into a shared atomic; checks that each g in 0..k-1 ran
    // exactly once per round, whatever k was in the round before (the pool keeps its workers between calls)
    if (k < 1 || k > kMaxDevices || rounds < 1) { set_error("selftest: k %d, rounds %d", k, rounds); return SMARTGPU_ERR_ARG; }
    for (int r = 0; r < rounds; ++r) {
        const int kk = 1 + (r * 7 + k - 1) % k;  // varies between 1 and k
        std::vector<int> ran(kk, 0);
        std::atomic<long> sum{0};
        g_pool.run(kk, [&](int g) { ++ran[g]; sum += g + 1; });
        long want = 0;
        for (int g = 0; g < kk; ++g) {
            want += g + 1;
            if (ran[g] != 1) { set_error("selftest: job %d of %d ran %d times in round %d", g, kk, ran[g], r); return SMARTGPU_ERR_HIP; }
        }
        if (sum != want) { set_error("selftest: sum %ld, want %ld", sum.load(), want); return SMARTGPU_ERR_HIP; }
    }
    return SMARTGPU_OK;
}

int smartgpu_msearch_batch64(int algo, const uint8_t* const* P, uint32_t m, uint32_t K, smartgpu_mtext* text, int reduce,
                             uint64_t* counts, double* pre_ms, double* batch_ms)
{
    if (!text) { set_error("text handle is NULL"); return SMARTGPU_ERR_ARG; }
    if (!P || K < 1 || !counts || !P[0]) { set_error("batch: P/counts NULL or K = 0"); return SMARTGPU_ERR_ARG; }
    if (K > kBatchMaxPatterns) { set_error("batch: %u patterns in one set (at most %u)", K, kBatchMaxPatterns); return SMARTGPU_ERR_ARG; }
    if (algo < 0 || algo >= SMARTGPU_NUM_ALGOS) { set_error("unknown algorithm id %d", algo); return SMARTGPU_ERR_ARG; }
    if (m < min_pattern(algo)) { set_error("%s: not applicable for m < %u", kAlgoNames[algo], min_pattern(algo)); return SMARTGPU_NA; }
    if (m < 1 || m > SMARTGPU_XSIZE) { set_error("pattern length %u outside [1,%d]", m, SMARTGPU_XSIZE); return SMARTGPU_ERR_ARG; }
    const int k = static_cast<int>(text->devices.size());
    // preprocessing: the K tables are built ONCE on the host and copied to the arena of every (distinct) device
    std::vector<BatchPlan> plans;
    std::vector<DeviceCtx*> ctx(k, nullptr), uniq;
    for (int g = 0; g < k; ++g) {
        ctx[g] = device_ctx(text->devices[g]);
        if (!ctx[g]) return SMARTGPU_ERR_HIP;
        if (std::find(uniq.begin(), uniq.end(), ctx[g]) == uniq.end()) uniq.push_back(ctx[g]);
    }
    const int up = batch_upload(uniq, algo, P, m, K, plans, pre_ms);
    if (up != SMARTGPU_OK) return up;
    const bool distinct = reduce == SMARTGPU_REDUCE_RCCL;
    if (distinct && static_cast<int>(uniq.size()) != k) { set_error("RCCL reduce: the %d devices of the text must be distinct", k); return SMARTGPU_ERR_ARG; }
    if (distinct && text->comms.empty()) {
        if (!g_rccl.load()) return SMARTGPU_ERR_HIP;
        text->comms.assign(k, nullptr);
        const ncclResult_t st = g_rccl.CommInitAll(text->comms.data(), k, text->devices.data());
        if (st != ncclSuccess) {
            set_error("ncclCommInitAll over %d devices failed: %s (devices must be distinct)", k, g_rccl.GetErrorString(st));
            text->comms.clear();
            return SMARTGPU_ERR_HIP;
        }
    }
    // searching: every shard on its own device / stream, concurrently; the same device listed twice (host reduce,
    // the one-GPU test of the shard arithmetic) shares one arena, so its shards take turns
    const double t_run = now_ms();
    std::vector<uint64_t> total(K, 0);
    auto launch_shard = [&](int g) -> int {
        DeviceCtx* d = ctx[g];
        HIP_TRY(hipSetDevice(text->devices[g]), return SMARTGPU_ERR_HIP);
        // shard g counts the starts it owns: its first (begin[g+1]-begin[g]) positions
        const uint64_t own = text->begin[g + 1] - text->begin[g];
        const uint64_t have = smartgpu_text_length(text->shards[g]);
        const uint64_t span = std::min<uint64_t>(have, own + m - 1);
        return batch_enqueue(d, algo, plans, m, text->shards[g], 0, span, K, false, nullptr);
    };
    int rc = SMARTGPU_OK;
    if (distinct) {
        // the k devices' launches are enqueued by k host threads at once (LaunchPool): each sets its device, enqueues the
        // memset and the kernel(s) of its shard; an error text travels back from the worker's thread-local slot
        std::vector<int> rcs(k, SMARTGPU_OK);
        std::vector<std::string> errs(k);
        g_pool.run(k, [&](int g) {
            rcs[g] = launch_shard(g);
            if (rcs[g] != SMARTGPU_OK) errs[g] = g_error;
        });
        for (int g = 0; g < k; ++g)
            if (rcs[g] != SMARTGPU_OK) { g_error = errs[g]; return rcs[g]; }
        // ONE collective for the whole pattern set: the K counts of every device, summed in place
        ncclResult_t st = g_rccl.GroupStart();
        for (int g = 0; g < k && st == ncclSuccess; ++g)
            st = g_rccl.AllReduce(ctx[g]->batch_counts, ctx[g]->batch_counts, K, ncclUint64, ncclSum, text->comms[g], ctx[g]->stream);
        const ncclResult_t end = g_rccl.GroupEnd();
        if (st == ncclSuccess) st = end;
        if (st != ncclSuccess) { set_error("RCCL all-reduce of %u counts over %d devices failed: %s", K, k, g_rccl.GetErrorString(st)); return SMARTGPU_ERR_HIP; }
        HIP_TRY(hipSetDevice(text->devices[0]), return SMARTGPU_ERR_HIP);
        HIP_TRY(hipMemcpyAsync(ctx[0]->pinned_counts, ctx[0]->batch_counts, static_cast<size_t>(K) * 8, hipMemcpyDeviceToHost, ctx[0]->stream),
                return SMARTGPU_ERR_HIP);
        HIP_TRY(hipStreamSynchronize(ctx[0]->stream), return SMARTGPU_ERR_HIP);
        for (uint32_t j = 0; j < K; ++j) total[j] = ctx[0]->pinned_counts[j];
        for (int g = 1; g < k; ++g) smartgpu_device_sync(text->devices[g]);
        for (uint32_t j = 0; j < K; ++j)
            if (count_poisoned(total[j])) return SMARTGPU_ERR_HIP;  // (a sum of at most 16 shards: still below 2^63)
    } else {
        for (int g = 0; g < k; ++g) {  // K-count read-backs added on the host
            rc = launch_shard(g);
            if (rc != SMARTGPU_OK) return rc;
            DeviceCtx* d = ctx[g];
            HIP_TRY(hipMemcpyAsync(d->pinned_counts, d->batch_counts, static_cast<size_t>(K) * 8, hipMemcpyDeviceToHost, d->stream),
                    return SMARTGPU_ERR_HIP);
            HIP_TRY(hipStreamSynchronize(d->stream), return SMARTGPU_ERR_HIP);
            for (uint32_t j = 0; j < K; ++j) {
                if (count_poisoned(d->pinned_counts[j])) return SMARTGPU_ERR_HIP;
                total[j] += d->pinned_counts[j];
            }
        }
    }
    const double run = now_ms() - t_run;
    for (uint32_t j = 0; j < K; ++j) counts[j] = total[j];
    if (batch_ms) *batch_ms = run;
    g_last_pre_ms = pre_ms ? pre_ms[K - 1] : 0.0;
    g_last_run_ms = run / K;
    return SMARTGPU_OK;
}

int smartgpu_msearch64(int algo, const uint8_t* P, uint32_t m, smartgpu_mtext* text, int reduce,
                       uint64_t* count, double* pre_ms, double* run_ms)
{
    if (!P) { set_error("pattern is NULL"); return SMARTGPU_ERR_ARG; }
    const uint8_t* one[1] = {P};
    uint64_t c = 0;
    double pre = 0.0, run = 0.0;
    const int rc = smartgpu_msearch_batch64(algo, one, m, 1, text, reduce, &c, &pre, &run);
    if (rc != SMARTGPU_OK) return rc;
    // (the per-device shards of the host-reduce path prepare their tables in turn: pre covers the last device)
    if (count) *count = c;
    if (pre_ms) *pre_ms = pre;
    if (run_ms) *run_ms = run;
    return SMARTGPU_OK;
}

/* ---- table export (tests) ---------------------------------------------- */
int smartgpu_build_table(int which, const uint8_t* P, uint32_t m, int32_t* out, uint32_t cap)
{
    if (!P || !out || m < 1 || m > SMARTGPU_XSIZE) { set_error("bad build_table arguments"); return SMARTGPU_ERR_ARG; }
    std::vector<int32_t> v;
    switch (which) {
        case 0: v = sg::bad_char(P, m); break;
        case 1: v = sg::good_suffix(P, m); break;
        case 2: v = sg::kmp_next(P, m); break;
        case 3: { auto s = sg::shift_or_masks(P, m); v.assign(s.begin(), s.end()); break; }
        case 4: { auto b = sg::bndm_masks(P, m); v.assign(b.begin(), b.end()); break; }
        case 5: {  // KMP transition table, (m+1)*256 entries (m <= 255)
            if (m > 255) { set_error("the KMP transition table needs m <= 255"); return SMARTGPU_ERR_ARG; }
            auto d = sg::kmp_dfa(P, m);
            v.assign(d.begin(), d.end());
            break;
        }
        case 6: {  // compressed form: [k1][colmap 256][table (m+1)*k1]
            if (m > 255) { set_error("the KMP transition table needs m <= 255"); return SMARTGPU_ERR_ARG; }
            uint32_t k1 = 0;
            auto d = sg::kmp_dfa_compressed(P, m, &k1);
            v.push_back(static_cast<int32_t>(k1));
            v.insert(v.end(), d.begin(), d.end());
            break;
        }
        case 7: { auto sa = sg::shift_and_masks(P, m); v.assign(sa.begin(), sa.end()); break; }
        case 9: {  // kmp_runs' tables as the kernel holds them in LDS (bytes; the last 272: Q and thr)
            const uint32_t w = sg::kmp_window(m);
            std::vector<uint8_t> t;
            sg::kmp_runs_tables(P, w, t);
            if (w < 63) {  // the blob stores the rows of the states one after the other: spread them out as the kernel does
                const uint32_t Z = 4 * w + 1;
                std::vector<uint8_t> lds((Z + 1) * 256 + 272, 0);
                for (uint32_t st = 0; st <= w; ++st) std::memcpy(&lds[4 * st * 256], &t[st * 256], 256);
                std::memset(&lds[Z * 256], static_cast<int>(Z), 256);
                std::memcpy(&lds[(Z + 1) * 256], &t[(w + 1) * 256], 272);
                t.swap(lds);
            }
            v.assign(t.begin(), t.end());
            break;
        }
        case 11: {  // kmp_runs' COMPACT tables as the kernel holds them in LDS: rows 0..w, row Z = 4(w+1), then Q and thr (272 bytes)
            const uint32_t w = sg::kmp_compact_window(m);
            std::vector<uint8_t> t;
            sg::kmp_runs_tables(P, w, t, true);
            const uint32_t Z = 4 * w + 4;
            std::vector<uint8_t> lds((w + 2) * 256 + 272, 0);
            std::memcpy(lds.data(), t.data(), (w + 1) * 256);
            std::memset(&lds[(w + 1) * 256], static_cast<int>(Z), 256);
            std::memcpy(&lds[(w + 2) * 256], &t[(w + 1) * 256], 272);
            v.assign(lds.begin(), lds.end());
            break;
        }
        case 10: {  // the two-bit codes of the byte values in P[0..m) taken as a SET (a text's alphabet): shift, symtab; no entries: none
            uint32_t set[8] = {0, 0, 0, 0, 0, 0, 0, 0}, shift = 7, symtab = 0;
            for (uint32_t i = 0; i < m; ++i) set[P[i] >> 5] |= 1u << (P[i] & 31);
            if (sg::four_symbol_codes(set, &shift, &symtab)) {
                v.push_back(static_cast<int32_t>(shift));
                v.push_back(static_cast<int32_t>(symtab));
            }
            break;
        }
        case 8: v = sg::quick_search_shifts(P, m); break;
        case 13: case 15: case 18: {  // HASHq: 256 shifts + the shift after a candidate
            const uint32_t q = static_cast<uint32_t>(which - 10);
            if (m < q) { set_error("HASH%u needs m >= %u", q, q); return SMARTGPU_ERR_ARG; }
            int32_t after = 1;
            v = sg::qgram_hash_shifts(P, m, q, &after);
            v.push_back(after);
            break;
        }
        default: set_error("unknown table %d", which); return SMARTGPU_ERR_ARG;
    }
    if (v.size() > cap) { set_error("table needs %zu entries, cap %u", v.size(), cap); return SMARTGPU_ERR_ARG; }
    std::memcpy(out, v.data(), v.size() * sizeof(int32_t));
    return static_cast<int>(v.size());
}

}  // extern "C"
