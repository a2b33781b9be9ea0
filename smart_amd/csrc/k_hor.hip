// k_hor.hip — the Horspool family on LDS tiles: hor_scan (HOR, TUNEDBM, RAITA, QS, HASHq), hor_scan_bp (KR)
// (one translation unit per kernel family: dev_common.hpp)
#include "dev_common.hpp"
#include "launch_common.hpp"

namespace sg {

// ---------------------------------------------------------------------------
// Horspool  (reference: src/algos/hor.c:26-51)
// LDS: u16 tab[256] | pattern tail P[m-1-H..m-1] | text [tile0-H16, tile0+TB)
// tab[c] = hbc[c] | 0x8000 when c == P[m-1]: the byte that selects the shift
// also answers the first comparison, so a window costs two LDS reads.
// ---------------------------------------------------------------------------
// VAR selects the member of the Horspool family — same tiles, same table
// layout, same verification machinery:
//   0  Horspool (hor.c) and Tuned BM (tunedbm.c:38-58: its zero table entry for P[m-1] and the
//      shift applied after a candidate are exactly the flag bit and the shift stored beside it;
//      its 3x-unrolled skip loop is this loop)
//   1  Raita (raita.c:52-60): Horspool's shifts; a candidate is tested last byte (the flag),
//      middle byte, first byte, then the rest — when the window is in LDS (m-1 <= halo); longer
//      windows are tested right to left through the halo and completed in memory as in 0
//   2  Quick Search (qs.c:27-52): the shift comes from the byte AFTER the window, T[s+m]; the
//      tile carries 16 more bytes at its end for it
//   9  Horspool again, the flat form for patterns whose symbols repeat (hor_flat above)
//   3, 5, 8  Lecroq's HASHq (hash3.c:28-84, hash5.c, hash8.c): the table is indexed by an 8-bit hash
//      of the window's last q = VAR bytes, h = sum T[e-k] * 2^k mod 256; its zero entry (the hash
//      of the pattern's last q-gram) is the flag, stored with the shift applied after a candidate
// Horspool, the flat form (VAR = 9; hor.c:33-51) for patterns whose symbols repeat (a.sparse == 0: natural language,
// medium alphabets): where windows survive their first comparison the loop below makes a wave wait for its one lane
// that walks.  As bm_scan: ONE loop over (e, k, sh) on lane tiles — a text byte c = T[e-k], the pattern byte P[m-1-k]
// and bc[c] per iteration; k = 0 opens a window and takes hbc[T[e]] with it (hor.c:49: the shift is always the LAST
// byte's), equal bytes walk on, the first unequal one — or the H+1-th equal one: an occurrence, or (LONG) a candidate
// for memory — moves the window.  A pure streaming scan (rand128: the headline) keeps the loop below: two LDS reads
// per window and a cheaper tile.
// LDS: u16 bc[256] | u8 ptail[32] (ptail[k] = P[m-1-k]) | lane tile (LaneTile<kBmHalo>)
template <int THREADS, int L, bool LONG>
__device__ __forceinline__ void hor_flat(const ScanArgs& a, uint64_t tile_first, uint32_t ntiles, uint8_t* smem)
{
    constexpr int TB = THREADS * L;
    using LT = LaneTile<kBmHalo>;
    static_assert(L == 64 && kBmHalo >= kHaloMax, "a lane owns one 64-byte segment of a lane tile");
    const uint32_t m = a.m, H = a.halo;
    uint16_t* bc = reinterpret_cast<uint16_t*>(smem);
    uint8_t* ptail = smem + 512;
    constexpr uint32_t kTxt = 512 + 32;
    uint8_t* txt = smem + kTxt;
    for (uint32_t i = threadIdx.x; i < 256; i += THREADS) bc[i] = reinterpret_cast<const uint16_t*>(a.blob + kTableOff)[i] & 0x7FFFu;
    for (uint32_t k = threadIdx.x; k < 32; k += THREADS) ptail[k] = k <= H ? a.blob[m - 1 - k] : 0;
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
        uint32_t x0 = 0, x1 = L;  // window ends [x0, x1) of the lane's segment are its own
        const uint64_t seg = tile0 + (uint64_t)threadIdx.x * L;
        if (tile0 < e_begin || tile0 + TB > e_end) {  // (uniform) a tile at either end of the range
            const uint64_t lo = seg > e_begin ? seg : e_begin;
            const uint64_t hi = seg + L < e_end ? seg + L : e_end;
            x0 = lo < hi ? (uint32_t)(lo - seg) : 0u;
            x1 = lo < hi ? (uint32_t)(hi - seg) : 0u;
        }
        // The lane's walk over its window ends.  HOW says what a window costs whose every byte in the tile is equal: 0 (the
        // whole window is in the tile) it is an occurrence; 1 (LONG, the walk every tile takes) it is counted and its end
        // remembered — Horspool's shift does not depend on where a window fails (hor.c:49), so the walk goes on at once and
        // the rest of the window, in HBM, is compared ONCE per tile instead of being tested for on every iteration
        // (bndm_scan); 2 (LONG, the lanes that saw more than one in this tile) the rest
        // is compared on the spot.
        uint32_t nocc = 0, last = 0;
        auto walk = [&](auto how) {
            constexpr int HOW = decltype(how)::value;
            uint32_t e = own + x0, k = 0, sh = 0;
            const uint32_t ehi = own + x1;
            while (e < ehi) {
                const uint32_t c = *(const lds_u8_t*)(size_t)(e - k);       // smem[e - k]
                const uint32_t pk = *(const lds_u8_t*)(size_t)(512u + k);   // ptail[k]
                const uint32_t t0 = *(const lds_u16_t*)(size_t)(2u * c);    // bc[c]
                sh = k == 0 ? t0 : sh;                 // hor.c:49: the shift is the window's LAST byte's
                const bool eq = c == pk;               // hor.c:46
                const bool full = eq && k == H;        // every byte the tile holds of the window is equal
                if (HOW == 0) {
                    hits += full;
                } else if (HOW == 1) {
                    nocc += full;
                    last = full ? e : last;
                } else if (full) {  // the window's first byte: text + seg + (e - own) - (m - 1)
                    hits += global_equal(a.text + seg + (e - own) - (m - 1), a.blob, m - 1 - H);
                }
                const bool on = eq && !full;
                e += on ? 0u : sh;
                k = on ? k + 1 : 0u;
            }
        };
        if (!LONG) {
            walk(std::integral_constant<int, 0>());
        } else {
            walk(std::integral_constant<int, 1>());
            if (__any(nocc != 0)) {  // rare, wave-uniform, once per tile
                if (nocc > 1) walk(std::integral_constant<int, 2>());
                hits += wave_verify(nocc == 1, a.text + seg + (last - own) - (m - 1), a.blob, m - 1 - H);
            }
        }
    }
    flush_hits(hits, a.count, smem, a.text);
}

template <int THREADS, int L, bool LONG, int VAR>  // LONG: m-1 > back halo, windows are completed in HBM
__global__ __launch_bounds__(THREADS) void hor_scan(ScanArgs a1, uint64_t tile_first,
                                                    uint32_t ntiles, const BatchItem* __restrict__ batch)
{
    const ScanArgs a = pick_args(a1, batch);  // a pattern set in one grid: blockIdx.y = pattern (launch_batch)
    constexpr int TB = THREADS * L;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    if constexpr (VAR == 9) {
        hor_flat<THREADS, L, LONG>(a, tile_first, ntiles, smem);
        return;
    }
    const uint32_t m = a.m, H = a.halo, H16 = round16(H);
    uint16_t* tab = reinterpret_cast<uint16_t*>(smem);
    uint8_t* ptail = smem + 512;                 // ptail[H-k] == P[m-1-k]
    uint8_t* txt = ptail + round16(H + 1);       // txt[H16 + x] == T[tile0 + x]

    for (uint32_t i = threadIdx.x; i < 256; i += THREADS)
        tab[i] = reinterpret_cast<const uint16_t*>(a.blob + kTableOff)[i];
    for (uint32_t i = threadIdx.x; i <= H; i += THREADS) ptail[i] = a.blob[m - 1 - H + i];

    const uint64_t e_begin = a.s_begin + m - 1, e_end = a.s_end + m - 1;
    uint32_t hits = 0;
    const uint64_t t_end = tile_first + ntiles;
    static_assert(TB == THREADS * 64, "prefetch registers are written out for L = 64");
    uint4 p0, p1, p2, p3, ph;  // prefetch registers: 4 tile rows + one halo chunk
    uint4 pf;                  // VAR 2: the 16 bytes after the tile (thread 0)
    const bool halo_lane = threadIdx.x * 16u < H16;
    const uint8_t plast = a.blob[m - 1];
    auto issue = [&](uint64_t tile0) {
        const uint8_t* src = a.text + tile0 + threadIdx.x * 16u;
        p0 = ld_stream16(src);
        p1 = ld_stream16(src + THREADS * 16);
        p2 = ld_stream16(src + THREADS * 32);
        p3 = ld_stream16(src + THREADS * 48);
        if (halo_lane) ph = ld_stream16(src - H16);
        if (VAR == 2 && threadIdx.x == 0) pf = *reinterpret_cast<const uint4*>(a.text + tile0 + TB);  // read again as the next tile's first row
    };
    uint64_t t = tile_first + blockIdx.x;
    issue(t * TB);
    for (; t < t_end; t += gridDim.x) {
        const uint64_t tile0 = t * TB;
        __syncthreads();  // previous tile fully consumed (and tables visible)
        {
            const uint32_t i0 = H16 + threadIdx.x * 16u;
            tile_park(txt, i0, p0);
            tile_park(txt, i0 + THREADS * 16, p1);
            tile_park(txt, i0 + THREADS * 32, p2);
            tile_park(txt, i0 + THREADS * 48, p3);
            if (halo_lane) tile_park(txt, threadIdx.x * 16u, ph);
            if (VAR == 2 && threadIdx.x == 0) tile_park(txt, H16 + TB, pf);
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
                uint32_t ent;
                if (VAR >= 3) {
                    uint32_t h = 0;
#pragma unroll
                    for (int q = VAR - 1; q >= 0; --q) h = (h << 1) + txt[tile_at(e - q)];
                    ent = tab[h & 0xFFu];
                } else {
                    ent = VAR == 2 ? tab[txt[tile_at(e + 1)]] : tab[txt[tile_at(e)]];
                    // (bm_scan's trick — reading the byte before the last along with it, so that a window
                    // that dies there costs no walk — measured here: m >= 32 unchanged, m = 8..16 74-84 % ->
                    // 48-74 %: this loop is LDS-bound at small m and the flat tile's byte reads conflict.)
                }
                if (VAR == 2 ? txt[tile_at(e)] == plast : (ent & 0x8000u) != 0) {
                    uint32_t k = VAR >= 3 ? 0 : 1;  // bytes matched so far, right to left (a hash proves nothing)
                    if (VAR == 1 && !LONG) {  // raita.c:56-57: middle byte, first byte
                        const uint32_t mid = m / 2;
                        if (txt[tile_at(e - (m - 1) + mid)] != ptail[H - (m - 1 - mid)] || txt[tile_at(e - (m - 1))] != ptail[H - (m - 1)])
                            k = H + 2;  // not a match
                    }
                    while (k <= H && ptail[H - k] == txt[tile_at(e - k)]) ++k;
                    bool ok = k == H + 1;
                    if (LONG && ok) {  // the rest of the window is not in LDS
                        const uint8_t* rest = a.text + tile0 + (e - H16) - (m - 1);
                        if (!parked) {
                            parked = true;
                            parked_at = rest;
                            ok = false;  // counted by wave_verify below
                        } else {
                            ok = global_equal(rest, a.blob, m - 1 - H);
                        }
                    }
                    hits += ok;
                }
                e += ent & 0x7FFFu;
            }
        }
        if (LONG) hits += wave_verify(parked, parked_at, a.blob, m - 1 - H);
    }
    flush_hits(hits, a.count, smem, a.text);
}

// ---------------------------------------------------------------------------
// Horspool, bank-private LDS layout (m <= 255).
//
// hor_scan above is LDS-bound for short patterns: rocprofv3 shows the LDS busy
// 95 % of the kernel at m=4 with 78 % of those cycles bank conflicts (lanes
// read random bytes of a flat tile and random entries of one shared table).
// Here every lane reads only its OWN bank:
//   * text: the 64-byte run of lane q is stored down column q of a
//     [16 rows][288 columns] dword matrix (row stride 1152 B = 9 * 128 B, so the
//     bank of a dword is column % 32 = q % 32 whatever the row).  Columns 0..31
//     hold the back halo (only the last ceil(H/64) are filled), 32+q is lane q.
//   * table: 32 copies of the 256-entry u8 shift table, copy b wholly in bank b
//     (8 KiB); lane l reads copy l % 32.
// Lanes l and l+32 of a wave share a bank but sit in different halves of the
// wave64 access, so reads are conflict-free by construction.
// The transpose happens on the way in: each wave loads its 4 KiB coalesced
// (4 x global_load_dwordx4), then lane (quad, r) writes register (r+t)%4 in
// step t, so the four lanes of a quad — whose registers hold the same four
// segments — hit two banks twice (a 2-way ds_write_b32 conflict is free,
// MI355X_MICROARCH.md §LDS) instead of one bank four times.
// ---------------------------------------------------------------------------
constexpr int kBpThreads = 256, kBpL = 64;
constexpr int kBpHaloCols = 32;                        // columns reserved for the back halo
constexpr int kBpCols = kBpHaloCols + kBpThreads;      // 288
constexpr int kBpRowBytes = kBpCols * 4;               // 1152
constexpr int kBpTextBytes = (kBpL / 4) * kBpRowBytes; // 18432
constexpr int kBpHB = kBpHaloCols * kBpL;              // P-coordinate of tile-local byte 0
constexpr int kBpTabBytes = 8192;

// LDS byte address of P-coordinate `P` (P = kBpHB + tile-local offset)
__device__ __forceinline__ uint32_t bp_addr(uint32_t P)
{
    return ((P & 63u) >> 2) * kBpRowBytes + (P >> 6) * 4u + (P & 3u);
}

// KR = true: Karp-Rabin (kr.c:26-54) on the same transposed tiles.  Every lane rolls a hash over
// its own 64 window ends, one byte in and (m < 32) one byte out per step, all lanes in lockstep:
// exactly the access pattern that would be a 32-way bank conflict on a flat tile and is conflict-
// free here, where lane q's bytes live in bank q.  The hash is the reference's: 32-bit, weights
// 2^(m-1-i), so of a window longer than 32 bytes only the last 32 still count and nothing has to
// be subtracted; an equal hash is confirmed byte by byte (LDS through the halo, then memory).
template <bool KR>
__global__ __launch_bounds__(kBpThreads) void hor_scan_bp(ScanArgs a1, uint64_t tile_first,
                                                          uint32_t ntiles, const BatchItem* __restrict__ batch)
{
    const ScanArgs a = pick_args(a1, batch);  // a pattern set in one grid: blockIdx.y = pattern (launch_batch)
    constexpr int TB = kBpThreads * kBpL;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t m = a.m, H = a.halo, H16 = round16(H);
    uint8_t* ptab = smem;                          // bank-private u8 shift tables
    uint8_t* ptail = smem + kBpTabBytes;           // ptail[H-k] == P[m-1-k]
    uint8_t* txt = ptail + round16(H + 1);
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;

    if (!KR) {  // 64 rows (4 chars each) x 32 banks: every bank gets the same packed dword
        const uint32_t* g = reinterpret_cast<const uint32_t*>(a.blob + kTableOff + 512);
        uint32_t* t32 = reinterpret_cast<uint32_t*>(ptab);
        for (uint32_t i = tid; i < 2048; i += kBpThreads) t32[i] = g[i >> 5];
    }
    const uint32_t kr_w = m < 32 ? m : 32;                                              // bytes the hash still sees
    const uint32_t kr_hp = KR ? *reinterpret_cast<const uint32_t*>(a.blob + kTableOff) : 0u;  // the pattern's hash
    for (uint32_t i = tid; i <= H; i += kBpThreads) ptail[i] = a.blob[m - 1 - H + i];
    const uint32_t plast = a.blob[m - 1];
    const uint32_t my_tab = (lane & 31u) * 4u;

    const uint64_t e_begin = a.s_begin + m - 1, e_end = a.s_end + m - 1;
    uint32_t hits = 0;
    const uint64_t t_end = tile_first + ntiles;
    const uint32_t r = lane & 3u;
    uint32_t* t32 = reinterpret_cast<uint32_t*>(txt);
    // prefetch registers: this wave's 4 KiB of the tile (coalesced) + one halo chunk
    uint4 v0, v1, v2, v3, hv;
    auto issue = [&](uint64_t tile0) {
        const uint8_t* src = a.text + tile0 + wave * 4096u + lane * 16u;
        v0 = ld_stream16(src);
        v1 = ld_stream16(src + 1024);
        v2 = ld_stream16(src + 2048);
        v3 = ld_stream16(src + 3072);
        if (tid * 16u < H16) hv = ld_stream16(a.text + tile0 - H16 + tid * 16u);
    };
    uint64_t t = tile_first + blockIdx.x;
    issue(t * TB);
    for (; t < t_end; t += gridDim.x) {
        const uint64_t tile0 = t * TB;
        __syncthreads();
        {   // ---- transposing stores of the prefetched tile
#pragma unroll
            for (uint32_t st = 0; st < 4; ++st) {
                const uint32_t k = (r + st) & 3u;
                uint4 x;
                x.x = k == 0 ? v0.x : k == 1 ? v1.x : k == 2 ? v2.x : v3.x;
                x.y = k == 0 ? v0.y : k == 1 ? v1.y : k == 2 ? v2.y : v3.y;
                x.z = k == 0 ? v0.z : k == 1 ? v1.z : k == 2 ? v2.z : v3.z;
                x.w = k == 0 ? v0.w : k == 1 ? v1.w : k == 2 ? v2.w : v3.w;
                const uint32_t col = kBpHaloCols + wave * 64u + k * 16u + (lane >> 2);
                const uint32_t w0 = (4u * r) * kBpCols + col;  // dword index of row 4r
                t32[w0] = x.x;
                t32[w0 + kBpCols] = x.y;
                t32[w0 + 2 * kBpCols] = x.z;
                t32[w0 + 3 * kBpCols] = x.w;
            }
            if (tid * 16u < H16) {  // back halo: bytes [tile0-H16, tile0)
                const uint32_t P0 = kBpHB - H16 + tid * 16u;
                const uint32_t w0 = ((P0 & 63u) >> 2) * kBpCols + (P0 >> 6);
                t32[w0] = hv.x;
                t32[w0 + kBpCols] = hv.y;
                t32[w0 + 2 * kBpCols] = hv.z;
                t32[w0 + 3 * kBpCols] = hv.w;
            }
        }
        __syncthreads();
        if (t + gridDim.x < t_end) issue((t + gridDim.x) * TB);
        const uint64_t seg = tile0 + (uint64_t)tid * kBpL;
        const uint64_t lo = seg > e_begin ? seg : e_begin;
        const uint64_t hi = seg + kBpL < e_end ? seg + kBpL : e_end;
        bool parked = false;  // KR, m-1 > H: first window of this tile whose hash and last H+1 bytes matched
        const uint8_t* parked_at = a.text;
        // the rest of such a window is in memory: the lane parks it for wave_verify below (64 lanes
        // compare 1 KiB per step) — done by the lane itself, one m=4096 occurrence cost 0.25 ms
        auto confirm_rest = [&](uint32_t ee) -> bool {
            const uint8_t* rest = a.text + tile0 + (ee - kBpHB) - (m - 1);
            if (!parked) {
                parked = true;
                parked_at = rest;
                return false;  // counted by wave_verify
            }
            return global_equal(rest, a.blob, m - 1 - H);
        };
        if (lo < hi) {
            uint32_t e = (uint32_t)(lo - tile0) + kBpHB;  // P-coordinates
            const uint32_t ehi = (uint32_t)(hi - tile0) + kBpHB;
            if (KR && m >= 8 && lo == seg && hi == seg + kBpL) {
                // A whole lane: the 64 window ends are this lane's own column, 16 dwords at compile-time
                // offsets, and the hash rolled is always the one of the last 32 bytes, from which nothing
                // has to be subtracted: a step is shift + add.  For m >= 32 that IS the reference's hash;
                // for m < 32 its low m bits are the low m bits of the reference's (bytes further back only
                // reach bits >= m), so those are compared — a filter of 2^-m instead of 2^-32, exact all
                // the same because every equal hash is confirmed byte by byte (m < 8: the rolling form
                // with the outgoing byte below).  Equal hashes are rare: a running minimum of the masked
                // difference says whether a group of 16 has one, and only then are its ends looked at.
                // Four bytes at a time: after the dword d = b0 b1 b2 b3 the hash is 16h + 8b0 + 4b1 + 2b2 + b3, and
                // the hashes at the three ends in between are (h << s) + the dot product of d with (2^(s-1), ..,
                // 1, 0, ..): one v_dot4_u32_u8 and one v_lshl_add each, all four from the same h (no chain
                // through the dword).  The hash the lane starts with — the 32 bytes before its first end — is
                // the lower half of the PREVIOUS lane's column: 8 dword reads, 8 dot products.
                const uint32_t kr_mask = kr_w == 32 ? 0xFFFFFFFFu : (1u << kr_w) - 1u;
                const uint8_t* col = txt + (kBpHaloCols + tid) * 4u;
                uint32_t h = 0;
#pragma unroll
                for (int r = 8; r < 16; ++r) {
                    const uint32_t d = *reinterpret_cast<const uint32_t*>(col - 4 + r * kBpRowBytes);
                    h = __builtin_amdgcn_udot4(d, 0x01020408u, h << 4, false);
                }
#pragma unroll 1
                for (uint32_t g = 0; g < 4; ++g) {
                    const uint32_t h0 = h;
                    uint32_t d[4], near = 0xFFFFFFFFu;
#pragma unroll
                    for (int r = 0; r < 4; ++r) d[r] = *reinterpret_cast<const uint32_t*>(col + (4 * g + r) * kBpRowBytes);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const uint32_t h1 = (h << 1) + __builtin_amdgcn_udot4(d[r], 0x00000001u, 0u, false);  // kr.c:26,48
                        const uint32_t h2 = (h << 2) + __builtin_amdgcn_udot4(d[r], 0x00000102u, 0u, false);
                        const uint32_t h3 = (h << 3) + __builtin_amdgcn_udot4(d[r], 0x00010204u, 0u, false);
                        h = (h << 4) + __builtin_amdgcn_udot4(d[r], 0x01020408u, 0u, false);
                        near = min(near, min((h1 ^ kr_hp) & kr_mask, (h2 ^ kr_hp) & kr_mask));
                        near = min(near, min((h3 ^ kr_hp) & kr_mask, (h ^ kr_hp) & kr_mask));
                    }
                    if (near == 0) {  // kr.c:47: some end in this group has the pattern's hash: which, from the registers
                        uint32_t hh = h0, hm = 0;
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            hh = (hh << 1) + ((d[i >> 2] >> (8 * (i & 3))) & 0xFFu);
                            hm |= ((hh ^ kr_hp) & kr_mask) == 0 ? (1u << i) : 0u;
                        }
                        while (hm) {  // ... and confirm those byte by byte
                            const uint32_t ee = e + 16 * g + (uint32_t)__builtin_ctz(hm);
                            hm &= hm - 1;
                            uint32_t k = 0;
                            while (k <= H && ptail[H - k] == txt[bp_addr(ee - k)]) ++k;
                            bool ok = k == H + 1;
                            if (ok && m - 1 > H) ok = confirm_rest(ee);
                            hits += ok;
                        }
                    }
                }
            } else if (KR) {
                uint32_t h = 0;  // hash of the window ending at e (kr.c:38-41)
                for (uint32_t k = 0; k < kr_w; ++k) h += (uint32_t)txt[bp_addr(e - k)] << k;
                for (;;) {
                    if (h == kr_hp) {  // kr.c:47: confirm
                        uint32_t k = 0;
                        while (k <= H && ptail[H - k] == txt[bp_addr(e - k)]) ++k;
                        bool ok = k == H + 1;
                        if (ok && m - 1 > H) ok = confirm_rest(e);
                        hits += ok;
                    }
                    if (++e >= ehi) break;
                    h = (h << 1) + txt[bp_addr(e)];                                  // kr.c:26,48: one byte in ...
                    if (kr_w < 32) h -= (uint32_t)txt[bp_addr(e - kr_w)] << kr_w;  // ... one byte out (weight 2^m)
                }
            } else {
                while (e < ehi) {
                    const uint32_t c = txt[bp_addr(e)];
                    const uint32_t shift = ptab[(c >> 2) * 128u + my_tab + (c & 3u)];
                    if (c == plast) {
                        uint32_t k = 1;
                        while (k <= H && ptail[H - k] == txt[bp_addr(e - k)]) ++k;
                        hits += k > H;  // m-1 == H here (m <= 255): the whole window was compared
                    }
                    e += shift;
                }
            }
        }
        if (KR && m - 1 > H) hits += wave_verify(parked, parked_at, a.blob, m - 1 - H);
    }
    flush_hits(hits, a.count, smem, a.text);
}


// ---------------------------------------------------------------------------
// launchers (the packed regime of short / repetitive patterns is chosen by the dispatcher, launch.hip)
// ---------------------------------------------------------------------------
hipError_t launch_hor(const ScanArgs& a, uint32_t q, int num_cus, hipStream_t stream)
{
    // Horspool's q-gram bad-character table (patterns over two to four symbols, api.cpp build_blob): hor_scan's HASHq loop
    if (q == 5) return launch_hor_var(SMARTGPU_HASH5, a, num_cus, stream);
    if (q == 8) return launch_hor_var(SMARTGPU_HASH8, a, num_cus, stream);
    const uint32_t m = a.m, H = a.halo;
    if (!a.sparse && m >= 2 && g_tune[2] != 3) {  // windows survive: the flat form, two-wave workgroups as bm_scan (tune(2,3): round 2's loop)
        const size_t flds = 512 + 32 + LaneTile<kBmHalo>::bytes(kBmBusyT);
        const TileRange ftr = tiles_for(a.s_begin + m - 1, a.s_end + m - 1, (uint64_t)kBmBusyT * kHorL);
        if (m - 1 > H) return launch_tiled(hor_scan<kBmBusyT, kHorL, true, 9>, a, ftr, kBmBusyT, flds, 12, num_cus, stream);
        return launch_tiled(hor_scan<kBmBusyT, kHorL, false, 9>, a, ftr, kBmBusyT, flds, 12, num_cus, stream);
    }
    const size_t lds = 512 + r16(H + 1) + ((r16(H) + (size_t)kHorT * kHorL + 16 + 63) & ~(size_t)63);  // whole 64-byte blocks: tile_at() permutes inside them
    const TileRange tr = tiles_for(a.s_begin + m - 1, a.s_end + m - 1, (uint64_t)kHorT * kHorL);
    if (m - 1 > H) return launch_tiled(hor_scan<kHorT, kHorL, true, 0>, a, tr, kHorT, lds, tile_wgs(a), num_cus, stream);
    return launch_tiled(hor_scan<kHorT, kHorL, false, 0>, a, tr, kHorT, lds, tile_wgs(a), num_cus, stream);
}

// the Horspool family on hor_scan's tiles: RAITA (VAR 1), QS (2), HASH3/5/8 (3/5/8)
hipError_t launch_hor_var(int algo, const ScanArgs& a, int num_cus, hipStream_t stream)
{
    const uint32_t m = a.m, H = a.halo;
    const size_t lds = 512 + r16(H + 1) + ((r16(H) + (size_t)kHorT * kHorL + 16 + 63) & ~(size_t)63);  // whole 64-byte blocks: tile_at() permutes inside them
    const TileRange tr = tiles_for(a.s_begin + m - 1, a.s_end + m - 1, (uint64_t)kHorT * kHorL);
#define SG_HOR_VAR(V_)                                                                                     \
    do {                                                                                                  \
        if (m - 1 > H) return launch_tiled(hor_scan<kHorT, kHorL, true, V_>, a, tr, kHorT, lds, tile_wgs(a), num_cus, stream); \
        return launch_tiled(hor_scan<kHorT, kHorL, false, V_>, a, tr, kHorT, lds, tile_wgs(a), num_cus, stream);    \
    } while (0)
    if (algo == SMARTGPU_QS) SG_HOR_VAR(2);
    if (algo == SMARTGPU_HASH3) SG_HOR_VAR(3);
    if (algo == SMARTGPU_HASH5) SG_HOR_VAR(5);
    if (algo == SMARTGPU_HASH8) SG_HOR_VAR(8);
    SG_HOR_VAR(1);
#undef SG_HOR_VAR
}

// Karp-Rabin: rolling hash on the bank-private tiles; a.halo = min(m-1, 32) (api.cpp)
hipError_t launch_kr(const ScanArgs& a, int num_cus, hipStream_t stream)
{
    const uint32_t m = a.m, H = a.halo;
    const size_t lds = kBpTabBytes + r16(H + 1) + kBpTextBytes;
    const TileRange tr = tiles_for(a.s_begin + m - 1, a.s_end + m - 1, (uint64_t)kBpThreads * kBpL);
    return launch_tiled(hor_scan_bp<true>, a, tr, kBpThreads, lds, 6, num_cus, stream);
}

#ifdef SMARTGPU_AB
// Horspool on the bank-private tiles (tune(0,2)); the product uses that kernel for Karp-Rabin only
hipError_t launch_hor_bp(const ScanArgs& a, int num_cus, hipStream_t stream)
{
    const uint32_t m = a.m, H = a.halo;
    const size_t lds = kBpTabBytes + r16(H + 1) + kBpTextBytes;
    const TileRange tr = tiles_for(a.s_begin + m - 1, a.s_end + m - 1, (uint64_t)kBpThreads * kBpL);
    return launch_tiled(hor_scan_bp<false>, a, tr, kBpThreads, lds, 5, num_cus, stream);
}
#endif


}  // namespace sg
