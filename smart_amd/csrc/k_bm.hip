// k_bm.hip — Boyer-Moore on lane tiles: bm_scan
// (one translation unit per kernel family: dev_common.hpp)
#include "dev_common.hpp"
#include "launch_common.hpp"

namespace sg {

// ---------------------------------------------------------------------------
// Boyer-Moore  (reference: src/algos/bm.c:27-93)
// LDS: u16 bc[256] | u32 walk[H+2] | lane tile (LaneTile<kBmHalo>)
//
// The lane loop is ONE flat loop over (e, k) — window end, bytes of the window matched so far — in which every
// iteration reads ONE text byte c = T[e-k], its bad-character entry bc[c] and walk[k] = (gs[m-1-k] << 9) | P[m-1-k]
// (whose address does not depend on the text: the two reads are in flight together), and does bm.c:83-89 for i = m-1-k:
//     c == P[m-1-k]  ->  k+1            else  ->  e += max(gs[m-1-k], bc[c] - k), k = 0
// A fresh window is nothing special: k = 0 compares with P[m-1] and shifts by max(gs[m-1], bc[c]).  An occurrence
// (m-1 = H: the whole window is in LDS) is the state k = H+1: walk[H+1] = (gs[0] << 9) | 0x100 never compares equal
// and moves on by gs[0] (bm.c:86; bc[.] - m <= 0), the lane counts it on the way.
// Round 2's loop opened a window with two folded tables and walked a surviving one in a nested loop: while one lane
// compared, the other 63 stood still, and every level of the nest was paid in exec-mask bookkeeping — 4.9 SCALAR
// instructions per text byte and lane next to 3.0 vector ones on English (m = 128, profiles/r03/b_pmc_bm_english_m128.txt:
// the CU's one scalar unit ~90 % busy).  Here a lane that compares and a lane that opens its next window run the
// same instructions; the only branches are the loop's own and, for long patterns, one wave-uniform test.
// ---------------------------------------------------------------------------
template <int THREADS, int L, bool LONG>  // LONG: m-1 > back halo
__global__ __launch_bounds__(THREADS) void bm_scan(ScanArgs a1, uint64_t tile_first,
                                                   uint32_t ntiles, const BatchItem* __restrict__ batch)
{
    const ScanArgs a = pick_args(a1, batch);  // a pattern set in one grid: blockIdx.y = pattern (launch_batch)
    constexpr int TB = THREADS * L;
    using LT = LaneTile<kBmHalo>;  // H <= kHaloMax = 16 bytes back, one more for the occurrence state (read, ignored)
    static_assert(L == 64 && kBmHalo >= kHaloMax + 1, "a lane owns one 64-byte segment of a lane tile");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t m = a.m, H = a.halo;
    uint16_t* bc = reinterpret_cast<uint16_t*>(smem);
    uint32_t* walk = reinterpret_cast<uint32_t*>(smem + 512);  // walk[k], 0 <= k <= H+1
    constexpr uint32_t kTxt = 512 + 4 * (kHaloMax + 2 + 2);    // 592: the lane tile
    uint8_t* txt = smem + kTxt;

    // the blob: u16 first[256], second[256] (round 2's folded tables; unused here), bc[256], gs[m], safe
    const uint16_t* gtab = reinterpret_cast<const uint16_t*>(a.blob + kTableOff);
    for (uint32_t i = threadIdx.x; i < 256; i += THREADS) bc[i] = gtab[512 + i];
    const uint32_t gs0 = gtab[768];       // bm.c:86: the shift after an occurrence
    const uint32_t safe = gtab[768 + m];  // for a parked window (api.cpp build_blob)
    for (uint32_t k = threadIdx.x; k <= H + 1; k += THREADS)
        walk[k] = k <= H ? ((uint32_t)gtab[768 + m - 1 - k] << 9) | a.blob[m - 1 - k] : (gs0 << 9) | 0x100u;
    if ((uint32_t)(uintptr_t)(lds_u8_t*)smem != 0u) {  // the walk below addresses LDS by offset
        if (threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(reinterpret_cast<unsigned long long*>(a.count), 1ull << 62);
        return;
    }

    const uint64_t e_begin = a.s_begin + m - 1, e_end = a.s_end + m - 1;
    uint32_t hits = 0;
    uint4 p0, p1, p2, p3, ph;  // prefetch registers: 4 tile rows + the 16 bytes in front of the tile
    auto issue = [&](uint64_t tile0) {
        const uint8_t* src = a.text + tile0 + threadIdx.x * 16u;
        p0 = ld_stream16(src);
        p1 = ld_stream16(src + THREADS * 16);
        p2 = ld_stream16(src + THREADS * 32);
        p3 = ld_stream16(src + THREADS * 48);
        if (threadIdx.x == 0) ph = ld_stream16(src - LT::DUP);
    };
    const uint64_t t_end = tile_first + ntiles;
    uint64_t t = tile_first + blockIdx.x;
    issue(t * TB);
    const uint32_t own = kTxt + threadIdx.x * LT::STRIDE + kBmHalo;  // LDS offset of the lane's own first byte
    for (; t < t_end; t += gridDim.x) {
        const uint64_t tile0 = t * TB;
        __syncthreads();
        LT::park(txt, threadIdx.x, p0, THREADS);
        LT::park(txt, THREADS + threadIdx.x, p1, THREADS);
        LT::park(txt, 2 * THREADS + threadIdx.x, p2, THREADS);
        LT::park(txt, 3 * THREADS + threadIdx.x, p3, THREADS);
        if (threadIdx.x == 0) LT::park_front(txt, 0, ph);
        __syncthreads();
        if (t + gridDim.x < t_end) issue((t + gridDim.x) * TB);
        // window ends [x0, x1) of the lane's segment are its own
        uint32_t x0 = 0, x1 = L;
        const uint64_t seg = tile0 + (uint64_t)threadIdx.x * L;
        if (tile0 < e_begin || tile0 + TB > e_end) {  // (uniform) a tile at either end of the range
            const uint64_t lo = seg > e_begin ? seg : e_begin;
            const uint64_t hi = seg + L < e_end ? seg + L : e_end;
            x0 = lo < hi ? (uint32_t)(lo - seg) : 0u;
            x1 = lo < hi ? (uint32_t)(hi - seg) : 0u;
        }
        // The lane's walk over its window ends.  HOW says what happens when the halo is exhausted (LONG: the H+1 bytes the
        // tile holds of a window are equal, the rest is in HBM): 1 (the walk every tile takes) the candidate is counted, its
        // end remembered, and the window moves on by a shift that is safe whatever the rest says (min of gs over the
        // positions still unchecked, from the host) — selects, no branch; the rest is compared ONCE per tile, after the
        // walk (hor_scan's flat form, bndm_scan: DESIGN.md section 4 round 3 item 9c); 2 (the lanes that saw more than one
        // candidate in this tile walk again) the rest is compared on the spot and the shift is bm.c:86/89's.  0: m-1 <= H,
        // the occurrence state k = H+1 counts.
        uint32_t nocc = 0, last = 0;
        auto walk_tile = [&](auto how) {
            constexpr int HOW = decltype(how)::value;
            uint32_t e = own + x0, k = 0;
            const uint32_t ehi = own + x1;
            while (e < ehi) {
                // m >= 2 here (launch_scan sends one-byte patterns to the packed matcher)
                const uint32_t c = *(const lds_u8_t*)(size_t)(e - k);          // smem[e - k]
                const uint32_t wk = *(const lds_u32_t*)(size_t)(512u + 4u * k);  // walk[k]
                const int b = (int)*(const lds_u16_t*)(size_t)(2u * c) - (int)k;  // bc[c] - k = bmBc[c] - m + 1 + i, i = m-1-k
                if (HOW == 0) hits += k > H;
                const bool eq = c == (wk & 0x1FFu);
                const int g = (int)(wk >> 9);
                uint32_t adv = eq ? 0u : (uint32_t)(g > b ? g : b);  // bm.c:89
                uint32_t nk = eq ? k + 1 : 0u;
                if (HOW == 1) {
                    const bool cand = nk > H;
                    nocc += cand;
                    last = cand ? e : last;
                    adv = cand ? safe : adv;
                    nk = cand ? 0u : nk;
                } else if (HOW == 2 && nk > H) {
                    const uint8_t* tp = a.text + seg + (e - own);  // the window's last byte
                    uint32_t kk = nk, cc = 0;
                    bool mismatch = false;
                    while (kk < m) {
                        cc = tp[-(int64_t)kk];
                        if (cc != a.blob[m - 1 - kk]) { mismatch = true; break; }
                        ++kk;
                    }
                    if (!mismatch) {
                        ++hits;
                        adv = gs0;
                    } else {
                        const int g2 = gtab[768 + m - 1 - kk], b2 = (int)bc[cc] - (int)kk;
                        adv = (uint32_t)(g2 > b2 ? g2 : b2);
                    }
                    nk = 0;
                }
                e += adv;
                k = nk;
                // (no lane leaves the loop in the occurrence state: the step that enters it does not move e)
            }
        };
        if (!LONG) {
            walk_tile(std::integral_constant<int, 0>());
        } else {
            walk_tile(std::integral_constant<int, 1>());
            if (__any(nocc != 0)) {  // rare, wave-uniform, once per tile
                if (nocc > 1) walk_tile(std::integral_constant<int, 2>());
                hits += wave_verify(nocc == 1, a.text + seg + (last - own) - (m - 1), a.blob, m - 1 - H);
            }
        }
    }
    flush_hits(hits, a.count, smem, a.text);
}


// ---------------------------------------------------------------------------
// launcher (m = 1, short and repetitive patterns: the dispatcher sends them to the packed matcher)
// ---------------------------------------------------------------------------
hipError_t launch_bm(const ScanArgs& a, int num_cus, hipStream_t stream)
{
    const uint32_t m = a.m, H = a.halo;
    // Patterns whose symbols repeat (a.sparse == 0: natural language, small alphabets): every lane is busy with
    // candidates and a workgroup waits for its slowest wave at each tile.  Two-wave workgroups, 12 per CU: English
    // m = 4 / 8 / 32 / 128: 55 / 71 / 73 / 72 % against 51 / 68 / 70 / 73 % with 6 four-wave workgroups (7: 52 / 63 / 69 / 70 %;
    // 14 two-wave: 52 / 64 / 66 / 69 %).
    if (g_tune[2] ? g_tune[2] == 2 : !a.sparse) {  // tune(2, 1 / 2): four-wave / two-wave workgroups
        const size_t lds = 512 + 4 * (kHaloMax + 2 + 2) + LaneTile<kBmHalo>::bytes(kBmBusyT);  // bc, walk, the lane tile
        const TileRange tr = tiles_for(a.s_begin + m - 1, a.s_end + m - 1, (uint64_t)kBmBusyT * kBmL);
        if (m - 1 > H) return launch_tiled(bm_scan<kBmBusyT, kBmL, true>, a, tr, kBmBusyT, lds, 12, num_cus, stream);
        return launch_tiled(bm_scan<kBmBusyT, kBmL, false>, a, tr, kBmBusyT, lds, 12, num_cus, stream);
    }
    const size_t lds = 512 + 4 * (kHaloMax + 2 + 2) + LaneTile<kBmHalo>::bytes(kBmT);  // bc, walk, the lane tile
    const TileRange tr = tiles_for(a.s_begin + m - 1, a.s_end + m - 1, (uint64_t)kBmT * kBmL);
    const int wgs = tile_wgs(a, true);
    if (m - 1 > H) return launch_tiled(bm_scan<kBmT, kBmL, true>, a, tr, kBmT, lds, wgs, num_cus, stream);
    return launch_tiled(bm_scan<kBmT, kBmL, false>, a, tr, kBmT, lds, wgs, num_cus, stream);
}


}  // namespace sg
