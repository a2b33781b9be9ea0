// k_bndmx.hip — the adjacent BNDM variants: sbndm_scan (SBNDM), bndml_scan (BNDML, multi-word)
// (one translation unit per kernel family: dev_common.hpp)
#include "dev_common.hpp"
#include "launch_common.hpp"

namespace sg {

// ---------------------------------------------------------------------------
// Simplified BNDM (sbndm.c:28-149) on round 2's BNDM tiles — flat, dword-swizzled, a nested loop per window (BNDM
// itself moved to bndm_scan above in round 3).  No bookkeeping of the longest prefix seen: a window that dies after k
// more bytes moves past the failing byte (shift w-k), an occurrence moves by the period of the (32-byte prefix of the)
// pattern, which the host stores after the fingerprint.
// w = min(m,32); tiles are indexed by the END of the w-byte (prefix) window.
// LDS: u32 B[256] | text [tile0-32, tile0+TB)
// ---------------------------------------------------------------------------
template <int THREADS, int L, bool LONG>  // LONG: m > 32, prefix hits are verified
__global__ __launch_bounds__(THREADS) void sbndm_scan(ScanArgs a1, uint64_t tile_first,
                                                     uint32_t ntiles, const BatchItem* __restrict__ batch)
{
    const ScanArgs a = pick_args(a1, batch);  // a pattern set in one grid: blockIdx.y = pattern (launch_batch)
    constexpr int TB = THREADS * L;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t m = a.m, w = m < 32 ? m : 32, H16 = 32;
    uint32_t* B = reinterpret_cast<uint32_t*>(smem);
    uint8_t* txt = smem + 1024;

    // masks left-aligned (B'[c] = B[c] << (32-w)): D <<= 1 then drops factors that can no
    // longer become a prefix, instead of carrying dead bits above bit w-1 as bndm.c's 32-bit
    // word does for m < 32 (they are cleared by the next AND either way: same D & B, same count)
    for (uint32_t i = threadIdx.x; i < 256; i += THREADS)
        B[i] = reinterpret_cast<const uint32_t*>(a.blob + kTableOff)[i] << (32 - w);

    const uint32_t period = *reinterpret_cast<const uint32_t*>(a.blob + kTableOff + 1024 + 32);
    const uint64_t e_begin = a.s_begin + w - 1, e_end = a.s_end + w - 1;
    uint32_t hits = 0;
    static_assert(TB == THREADS * 64, "prefetch registers are written out for L = 64");
    uint4 p0, p1, p2, p3, ph;  // prefetch registers: 4 tile rows + one halo chunk
    const bool halo_lane = threadIdx.x * 16u < H16;
    auto issue = [&](uint64_t tile0) {
        const uint8_t* src = a.text + tile0 + threadIdx.x * 16u;
        p0 = ld_stream16(src);
        p1 = ld_stream16(src + THREADS * 16);
        p2 = ld_stream16(src + THREADS * 32);
        p3 = ld_stream16(src + THREADS * 48);
        if (halo_lane) ph = ld_stream16(src - H16);
    };
    const uint64_t t_end = tile_first + ntiles;
    uint64_t t = tile_first + blockIdx.x;
    issue(t * TB);
    for (; t < t_end; t += gridDim.x) {
        const uint64_t tile0 = t * TB;
        __syncthreads();
        {   // dword-swizzled like hor_scan's tile (tile_at)
            const uint32_t i0 = H16 + threadIdx.x * 16u;
            tile_park(txt, i0, p0);
            tile_park(txt, i0 + THREADS * 16, p1);
            tile_park(txt, i0 + THREADS * 32, p2);
            tile_park(txt, i0 + THREADS * 48, p3);
            if (halo_lane) tile_park(txt, threadIdx.x * 16u, ph);
        }
        __syncthreads();
        if (t + gridDim.x < t_end) issue((t + gridDim.x) * TB);
        const uint64_t seg = tile0 + (uint64_t)threadIdx.x * L;
        const uint64_t lo = seg > e_begin ? seg : e_begin;
        const uint64_t hi = seg + L < e_end ? seg + L : e_end;
        bool parked = false;  // first candidate of this tile awaiting wave_verify
        const uint8_t* parked_at = a.text;
        if (lo < hi) {
            uint32_t e = (uint32_t)(lo - tile0) + H16;
            const uint32_t ehi = (uint32_t)(hi - tile0) + H16;
            while (e < ehi) {
                // bndm.c:49-58 with the first step peeled: D = ~0 & B[c], and B[c] == 0 (c does
                // not occur in the prefix) moves the window by w after one text and one table read
                uint32_t D = B[txt[tile_at(e)]];
                if (D == 0) {
                    // sbndm.c:60-63 reads a second byte before it tests D and so moves by w-1
                    // here; its long-pattern form skips by w like BNDM (sbndm.c:133)
                    e += !LONG ? w - 1 : w;
                    continue;
                }
                uint32_t k = 1;
                for (;;) {  // sbndm.c:61-65
                    D = (D << 1) & B[txt[tile_at(e - k)]];
                    if (k == w - 1 || D == 0) break;
                    ++k;
                }
                if (D != 0) {  // the whole window matched
                    if (!LONG) {
                        ++hits;
                    } else {
                        const uint8_t* rest = a.text + tile0 + (e - H16) + 1;  // = text + s + w
                        if (!parked) {
                            parked = true;
                            parked_at = rest;
                        } else {
                            hits += global_equal(rest, a.blob + w, m - w);
                        }
                    }
                    e += period;
                } else {
                    e += w - k;
                }
            }
        }
        if (LONG) hits += wave_verify(parked, parked_at, a.blob + w, m - w);
    }
    flush_hits(hits, a.count, smem, a.text);
}

// ---------------------------------------------------------------------------
// BNDM with multi-word bit vectors  (src/algos/bndml.c:82-132, search_large; m <= 32 is plain BNDM
// and runs on bndm_scan).  The whole window lives in W = 2, 4 or 8 words held in registers, the
// shift carries from word to word, bit w-1 of D after k bytes says "the last k bytes are a prefix
// of P" (shift = w - longest such k).  w = min(m, kBndmlWindow = 64), W = 2: the reference keeps
// ceil(m/32) words for any m (its table is 128 KB at m = 4096); here a longer pattern is filtered by
// its 64-byte prefix and the rest is verified in memory, as the single-word algorithms do with 32.
// (The kernel is written for any W; with 256-byte windows, W = 8, the eight-word shift per text byte
// made it VALU-bound — 72-80 % for m >= 256 against 82-85 % with two words — and a streaming scan has
// no use for shifts longer than a lane's 64 bytes.)
// Tiles are indexed by the END of the w-byte window with a 256-byte back halo.
// LDS: u32 B[256][W] | P[0..w) | text [tile0-256, tile0+TB)
// ---------------------------------------------------------------------------
template <int THREADS, int L, int W, bool LONG>  // LONG: m > kBndmlWindow
__global__ __launch_bounds__(THREADS) void bndml_scan(ScanArgs a1, uint64_t tile_first, uint32_t ntiles, const BatchItem* __restrict__ batch)
{
    const ScanArgs a = pick_args(a1, batch);  // a pattern set in one grid: blockIdx.y = pattern (launch_batch)
    constexpr int TB = THREADS * L;
    constexpr uint32_t H16 = 256;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t m = a.m, w = m < kBndmlWindow ? m : kBndmlWindow;
    uint32_t* B = reinterpret_cast<uint32_t*>(smem);
    uint8_t* pw = smem + 256 * W * 4;  // P[0..w), for the direct comparison below
    uint8_t* txt = pw + 256;
    for (uint32_t i = threadIdx.x; i < 256 * W; i += THREADS)
        B[i] = reinterpret_cast<const uint32_t*>(a.blob + kTableOff)[i];
    for (uint32_t i = threadIdx.x; i < w; i += THREADS) pw[i] = a.blob[i];
    // shift after an occurrence: the period of P[0..w) — what the walk below would find as w - longest
    const uint32_t period = *reinterpret_cast<const uint32_t*>(a.blob + kTableOff + 256 * W * 4);
    const uint32_t top_word = (w - 1) >> 5, top_bit = 1u << ((w - 1) & 31u);

    const uint64_t e_begin = a.s_begin + w - 1, e_end = a.s_end + w - 1;
    uint32_t hits = 0;
    static_assert(TB == THREADS * 64, "prefetch registers are written out for L = 64");
    uint4 p0, p1, p2, p3, ph;  // prefetch registers: 4 tile rows + one halo chunk
    const bool halo_lane = threadIdx.x * 16u < H16;
    auto issue = [&](uint64_t tile0) {
        const uint8_t* src = a.text + tile0 + threadIdx.x * 16u;
        p0 = ld_stream16(src);
        p1 = ld_stream16(src + THREADS * 16);
        p2 = ld_stream16(src + THREADS * 32);
        p3 = ld_stream16(src + THREADS * 48);
        if (halo_lane) ph = *reinterpret_cast<const uint4*>(src - H16);
    };
    const uint64_t t_end = tile_first + ntiles;
    uint64_t t = tile_first + blockIdx.x;
    issue(t * TB);
    for (; t < t_end; t += gridDim.x) {
        const uint64_t tile0 = t * TB;
        __syncthreads();
        {
            uint8_t* dst = txt + H16 + threadIdx.x * 16u;
            *reinterpret_cast<uint4*>(dst) = p0;
            *reinterpret_cast<uint4*>(dst + THREADS * 16) = p1;
            *reinterpret_cast<uint4*>(dst + THREADS * 32) = p2;
            *reinterpret_cast<uint4*>(dst + THREADS * 48) = p3;
            if (halo_lane) *reinterpret_cast<uint4*>(txt + threadIdx.x * 16u) = ph;
        }
        __syncthreads();
        if (t + gridDim.x < t_end) issue((t + gridDim.x) * TB);
        const uint64_t seg = tile0 + (uint64_t)threadIdx.x * L;
        const uint64_t lo = seg > e_begin ? seg : e_begin;
        const uint64_t hi = seg + L < e_end ? seg + L : e_end;
        bool parked = false;  // first candidate of this tile awaiting wave_verify
        const uint8_t* parked_at = a.text;
        if (lo < hi) {
            uint32_t e = (uint32_t)(lo - tile0) + H16;
            const uint32_t ehi = (uint32_t)(hi - tile0) + H16;
            while (e < ehi) {
                uint32_t D[W], alive = 0;
                {
                    const uint32_t* b = B + (uint32_t)txt[e] * W;  // bndml.c:100-103
#pragma unroll
                    for (int i = 0; i < W; ++i) { D[i] = b[i]; alive |= D[i]; }
                }
                if (alive == 0) {  // the byte does not occur in the (prefix of the) pattern
                    e += w;
                    continue;
                }
                uint32_t k = 1, longest = 0;
                while (k < w && alive != 0) {  // bndml.c:104-116
                    if (k == 32) {
                        // Still alive 32 bytes deep.  The walk is not continued: by ONE lane with W-word
                        // shifts it cost ~0.26 ms per occurrence (each is surrounded by windows that stay
                        // alive for up to w bytes).  D already says where these 32 bytes occur in P:
                        // bit b <=> they are P[w-1-b .. w-1-b+32), i.e. P would end r = b - 31 bytes to
                        // the right of this window.  The lowest set bit is the nearest such alignment
                        // and a safe shift — the one the full walk arrives at, too.  r = 0 (they are P's
                        // suffix) is settled by comparing the window with P directly.
                        auto lowest = [&]() -> uint32_t {  // index of the lowest set bit of D, or 32*W
                            uint32_t b = 32u * W;
#pragma unroll
                            for (int i = W - 1; i >= 0; --i) b = D[i] != 0 ? 32u * i + (uint32_t)__builtin_ctz(D[i]) : b;
                            return b;
                        };
                        uint32_t b = lowest();
                        if (b == 31) {
                            uint32_t j = 0;
                            while (j < w && txt[e - j] == pw[w - 1 - j]) ++j;
                            if (j == w) {  // an occurrence: count it below, move on by the period
                                k = w;
                                longest = w - period;
                                break;
                            }
#pragma unroll
                            for (int i = 0; i < W; ++i) D[i] = i == 0 ? (D[i] & 0x7FFFFFFFu) : D[i];
                            b = lowest();
                        }
                        const uint32_t r = b < 32u * W ? b - 31u : w;  // nearest remaining alignment
                        const uint32_t sh = r < w - longest ? r : w - longest;
                        longest = w - sh;
                        alive = 0;
                        break;
                    }
                    uint32_t top = 0;
#pragma unroll
                    for (int i = 0; i < W; ++i) top = (uint32_t)i == top_word ? D[i] : top;
                    if (top & top_bit) longest = k;
                    const uint32_t* b = B + (uint32_t)txt[e - k] * W;
                    uint32_t carry = 0;
                    alive = 0;
#pragma unroll
                    for (int i = 0; i < W; ++i) {
                        const uint32_t cur = D[i];
                        D[i] = ((cur << 1) | carry) & b[i];
                        carry = cur >> 31;
                        alive |= D[i];
                    }
                    ++k;
                }
                if (alive != 0) {  // all w bytes matched
                    if (!LONG) {
                        ++hits;
                    } else {
                        const uint8_t* rest = a.text + tile0 + (e - H16) + 1;  // = text + s + w
                        if (!parked) {
                            parked = true;
                            parked_at = rest;
                        } else {
                            hits += global_equal(rest, a.blob + w, m - w);
                        }
                    }
                }
                e += w - longest;  // bndml.c:118
            }
        }
        if (LONG) hits += wave_verify(parked, parked_at, a.blob + w, m - w);
    }
    flush_hits(hits, a.count, smem, a.text);
}


// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
hipError_t launch_sbndm(const ScanArgs& a, int num_cus, hipStream_t stream)
{
    const uint32_t m = a.m, w = m < 32 ? m : 32;
    const size_t lds = 1024 + ((32 + (size_t)kBndmT * kBndmL + 63) & ~(size_t)63);  // whole 64-byte blocks (tile_at)
    const TileRange tr = tiles_for(a.s_begin + w - 1, a.s_end + w - 1, (uint64_t)kBndmT * kBndmL);
    if (m > 32) return launch_tiled(sbndm_scan<kBndmT, kBndmL, true>, a, tr, kBndmT, lds, tile_wgs(a), num_cus, stream);
    return launch_tiled(sbndm_scan<kBndmT, kBndmL, false>, a, tr, kBndmT, lds, tile_wgs(a), num_cus, stream);
}

// multi-word vectors, m > 32 (m <= 32 is plain BNDM, bndml.c:44-75: launch_bndm)
hipError_t launch_bndml(const ScanArgs& a, int num_cus, hipStream_t stream)
{
    const uint32_t m = a.m, w = m < kBndmlWindow ? m : kBndmlWindow;
    const TileRange tr = tiles_for(a.s_begin + w - 1, a.s_end + w - 1, (uint64_t)kBndmT * kBndmL);
    static_assert(kBndmlWindow <= 64, "wider windows: instantiate bndml_scan with W = 4 (<= 128 bytes) or 8 (<= 256)");
    constexpr int W = 2;
    const size_t lds = 256 * W * 4 + 256 + 256 + (size_t)kBndmT * kBndmL;
    if (m > kBndmlWindow) return launch_tiled(bndml_scan<kBndmT, kBndmL, W, true>, a, tr, kBndmT, lds, tile_wgs(a), num_cus, stream);
    return launch_tiled(bndml_scan<kBndmT, kBndmL, W, false>, a, tr, kBndmT, lds, tile_wgs(a), num_cus, stream);
}


}  // namespace sg
