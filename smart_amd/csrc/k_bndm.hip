// k_bndm.hip — BNDM with q-grams on column tiles: bndm_scan
// (one translation unit per kernel family: dev_common.hpp)
#include "dev_common.hpp"
#include "launch_common.hpp"

namespace sg {

// ---------------------------------------------------------------------------
// BNDM on a TEXT of at most four distinct byte values (round 4; TextCodes: what a text consists of is known since it
// was created): the window's last Q bytes are one of 256 GRAMS — Q = 8 one-bit symbols on a text of two values (GRAM = 1),
// Q = 4 two-bit symbols on three or four (GRAM = 2) — and what bndm.c:44-60 computes from them is a function of the gram
// alone, tabulated by the workgroup before it starts (256 entries, Q steps of the recurrence each, from B and the text's
// codes):
//   * D after the Q steps lists every place i where the gram is a factor of P (bit 31 - i <=> gram == P[i..i+Q));
//   * the factor at i = w - Q is the window's own place: a CANDIDATE (the whole window if w == Q);
//   * the nearest factor to its left, i_max, is the smallest shift that can align the gram with P again: w - Q - i_max;
//     with no factor at all the shift is bndm.c:54's `last` — w minus the longest suffix of the gram that is a prefix of
//     P (the prefix hits of steps 1..Q-1) — or w.
// E[gram] = candidate << 31 | shift: ONE lookup per WINDOW — no state between windows, no second iteration: a window
// whose first gram is alive is not read further unless it is a candidate (then the w - Q bytes before the gram are
// compared with P[0..w-Q) in LDS; m > 32: the rest in memory, first candidate of a tile parked for wave_verify).  The
// mask loop below reads on while factors are alive and, forgetting what it saw inside a gram, moves a fully read
// window by 1 — the lanes that meet such windows set their wave's trip count (rand2 m = 16: 0.35, here the shift is ~14).
// LDS: u32 E[256] (first B: the masks, left-aligned) | column tile | P[0..32)
// ---------------------------------------------------------------------------
template <int THREADS, int L, bool LONG, int Q, int GRAM>
__device__ __forceinline__ void bndm_gram(const ScanArgs& a, uint64_t tile_first, uint32_t ntiles, uint8_t* smem)
{
    constexpr int TB = THREADS * L;
    using CT = ColTile<THREADS>;
    static_assert(L == 64 && THREADS == 256 && ((GRAM == 1 && Q == 8) || (GRAM == 2 && Q == 4)), "a gram is 8 one-bit or 4 two-bit symbols; thread g derives entry g");
    const uint32_t m = a.m, w = m < 32 ? m : 32;  // w >= Q (launch_bndm)
    uint32_t* E = reinterpret_cast<uint32_t*>(smem);
    // (the tile at a multiple of 256 — its rows are then ds_read2st64's immediate offsets, no add on the lane's address —, the
    // pattern's bytes BEHIND it: 1056 bytes + the tile as before, six workgroups per CU)
    constexpr uint32_t kTxt = 1024, kPat = kTxt + ColTile<THREADS>::bytes();
    uint8_t* txt = smem + kTxt;
    // the text's codes, from the first words of the text's own allocation (TextCodes, kernels.hpp)
    const uint32_t* const tc = reinterpret_cast<const uint32_t*>(a.text - kFrontPad);
    const uint32_t cshift = GRAM == 2 ? tc[0] : tc[2] & 0xFFu;  // two-bit codes: (c >> shift) & 3; one-bit: (c >> bit) & 1
    const uint32_t symtab = GRAM == 2 ? tc[1] : tc[2] >> 8;     // the byte value of each code
    E[threadIdx.x] = reinterpret_cast<const uint32_t*>(a.blob + kTableOff)[threadIdx.x] << (32 - w);  // B, left-aligned
    if (threadIdx.x < 8) reinterpret_cast<uint32_t*>(smem + kPat)[threadIdx.x] = reinterpret_cast<const uint32_t*>(a.blob)[threadIdx.x];
    if ((uint32_t)(uintptr_t)(lds_u8_t*)smem != 0u) {  // the walk below addresses LDS by offset
        if (threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(reinterpret_cast<unsigned long long*>(a.count), 1ull << 62);
        return;
    }
    __syncthreads();
    {
        const uint32_t g = threadIdx.x;
        uint32_t D = 0xFFFFFFFFu, part = 0;
#pragma unroll
        for (int j = 0; j < Q; ++j) {  // byte Q-1-j of the gram: the j-th byte bndm.c:50 reads (right to left)
            const int i = Q - 1 - j;
            const uint32_t code = GRAM == 2 ? (g >> (2 * i)) & 3u : (g >> i) & 1u;
            D &= E[(symtab >> (8 * code)) & 0xFFu];           // bndm.c:51
            if (j + 1 < Q) {
                if ((int32_t)D < 0) part = (uint32_t)j + 1;  // bndm.c:52-54: these j+1 bytes are a prefix of P
                D <<= 1;                                      // bndm.c:57
            }
        }
        const uint32_t own = 31u - (w - Q);                   // bit of the factor at i = w - Q
        const uint32_t cand = (D >> own) & 1u;
        const uint32_t left = own == 31u ? 0u : D >> (own + 1u) << (own + 1u);  // the factors at i < w - Q
        const uint32_t shift = left ? (w - Q) - (31u - (uint32_t)__builtin_ctz(left)) : w - part;
        __syncthreads();  // every thread has read B
        E[g] = (cand << 31) | shift;
    }

    const uint64_t e_begin = a.s_begin + w - 1, e_end = a.s_end + w - 1;
    uint32_t hits = 0;
    uint4 pre[4], ph;  // prefetch registers: 4 tile rows + (threads 0, 1) the 32 bytes in front of the tile
    auto issue = [&](uint64_t tile0) {
        const uint8_t* src = a.text + tile0 + threadIdx.x * 16u;
        pre[0] = ld_stream16(src);
        pre[1] = ld_stream16(src + THREADS * 16);
        pre[2] = ld_stream16(src + THREADS * 32);
        pre[3] = ld_stream16(src + THREADS * 48);
        if (threadIdx.x < 2) ph = ld_stream16(src - 32);
    };
    const uint64_t t_end = tile_first + ntiles;
    uint64_t t = tile_first + blockIdx.x;
    issue(t * TB);
    const uint32_t col4 = kTxt + CT::col(threadIdx.x) * 4u;  // the lane's column; position 32 + x = byte x of its segment
    // four text bytes from position p of the lane's column (two aligned dwords a row apart + v_alignbyte_b32)
    auto text4 = [&](uint32_t p) -> uint32_t {
        const uint32_t at = col4 + (p >> 2) * CT::RS;
        return __builtin_amdgcn_alignbyte(*(const lds_u32_t*)(size_t)(at + CT::RS), *(const lds_u32_t*)(size_t)at, p);
    };
    const uint32_t nrest = w - Q;  // bytes of the window in front of its last gram
    const uint32_t rot4 = (cshift + 30u) & 31u;  // GRAM 2: rotate right by shift - 2 (left by 2 - shift): a byte's code at its bits 2-3
    for (; t < t_end; t += gridDim.x) {
        const uint64_t tile0 = t * TB;
        __syncthreads();
        CT::park(txt, pre, ph);
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
        // The walk runs on pl = e - (Q - 1), the position of the gram's FIRST byte (e: the window's end): that is what addresses the
        // column and shifts the two dwords apart, and the loop is bound by what it issues — 14 vector instructions per window
        // before, 11 now (no e - 3, no + 32 for the tile's offset, the codes scaled by 4 as they are extracted: the dot product is
        // the table's byte offset).
        uint32_t pl = 32u + x0 - (Q - 1);
        const uint32_t plhi = 32u + x1 - (Q - 1);
        uint32_t parked_e = 0;  // LONG: the window end of the tile's first candidate whose 32 bytes matched (0: none; e >= 32)
        while (pl < plhi) {
            const uint32_t at = col4 + (pl >> 2) * CT::RS;
            const uint32_t w0 = *(const lds_u32_t*)(size_t)at;
            const uint32_t w1 = *(const lds_u32_t*)(size_t)(at + CT::RS);
            const uint32_t x_lo = __builtin_amdgcn_alignbyte(w1, w0, pl);
            uint32_t g4;  // 4 * gram: E's byte offset
            if (GRAM == 2) {
                // the two-bit codes at bits 2-3 of their bytes (a rotation by shift - 2, either way), weights 1, 4, 16, 64
                g4 = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbit(x_lo, x_lo, rot4) & 0x0C0C0C0Cu, 0x40100401u, 0u, false);
            } else {
                const uint32_t w2 = *(const lds_u32_t*)(size_t)(at + 2 * CT::RS);
                const uint32_t x_hi = __builtin_amdgcn_alignbyte(w2, w1, pl);
                g4 = __builtin_amdgcn_udot4((x_lo >> cshift) & 0x01010101u, 0x20100804u, 0u, false) |
                     (__builtin_amdgcn_udot4((x_hi >> cshift) & 0x01010101u, 0x08040201u, 0u, false) << 6);
            }
            const uint32_t ent = *(const lds_u32_t*)(size_t)g4;
            if (!LONG && nrest == 0) {  // (uniform) the window is one gram
                hits += ent >> 31;
            } else if (__any((int32_t)ent < 0) && (int32_t)ent < 0) {  // a candidate: one gram in 256 on random text
                // the nrest bytes in front of the gram against P[0..nrest), a dword at a time
                const uint32_t e = pl + (Q - 1);
                const uint32_t ws = e - (w - 1);
                bool ok = true;
                for (uint32_t d = 0; d < nrest; d += 4) {
                    const uint32_t nb = nrest - d < 4 ? nrest - d : 4u;
                    const uint32_t mask = nb == 4 ? 0xFFFFFFFFu : (1u << (8u * nb)) - 1u;
                    ok = ok && ((text4(ws + d) ^ *(const lds_u32_t*)(size_t)(kPat + d)) & mask) == 0;
                }
                if (ok) {
                    if (!LONG) ++hits;
                    else if (parked_e == 0) parked_e = e;
                    else hits += global_equal(a.text + seg + (e - 32u) + 1, a.blob + w, m - w);  // = text + s + w
                }
            }
            pl += ent & 0xFFu;
        }
        if (LONG && __any(parked_e != 0)) hits += wave_verify(parked_e != 0, a.text + seg + (parked_e - 32u) + 1, a.blob + w, m - w);
    }
    flush_hits(hits, a.count, smem, a.text);
}

// ---------------------------------------------------------------------------
// BNDM with q-grams, 32-bit words like the reference  (src/algos/bndm.c:27-111; reading q bytes of a window at
// once is bndmq2.c / bndmq4.c:29-72's idea).  w = min(m,32); tiles are indexed by the END of the w-byte (prefix) window.
// LDS: u32 B[256] (left-aligned: B[c] << (32-w)) | column tile (ColTile)
//
// One flat loop over (e, k, D) — window end, bytes of the window read, the factors of P still alive — in which every
// iteration reads the NEXT Q BYTES of the window, T[e-k-Q+1 .. e-k], with ONE unaligned LDS read, looks up their Q
// masks and takes Q steps of bndm.c:49-58 at once (bndmq4.c:29's GRAM4):
//     t = (D << (Q-1)) & (B[c_0] << (Q-1)) & (B[c_1] << (Q-2)) & ... & B[c_{Q-1}],   D' = t << 1
// (Q | w: a window is read through in whole iterations).  The sign bit of t <=> the k+Q bytes read are a prefix of P —
// all w of them: an occurrence.  D' == 0 — no factor alive, or the window read through (the masks are left-aligned: the
// last bit leaves with the w-th step) — ends the window, and e moves by w - (k+Q) + 1: the k+Q bytes are no factor of P
// (or all of it), the k+Q-1 after their first may be (bndmq4.c:61: i += m-q+1) — by w - (k+Q) when they are a prefix
// of P themselves (bndm.c:54; what it remembers INSIDE a gram — a longer safe shift now and then — is not kept: the
// masks of a gram are ANDed before anything is tested).
// Lanes that open a window and lanes that are deep in one run the same instructions; no nested loop, no divergence
// beyond the loop's own exit.  3Q + 14 VALU instructions and Q + 1 LDS reads per iteration.
// Q comes from the plan (api.cpp build_blob, from the pattern's own symbol statistics): the smallest of 1, 2, 4, 8 for
// which most windows die in their first iteration.  On a large alphabet that is 1 — one text byte, one mask, as round
// 2's loop; English: 2; four symbols: 4; two: 8 — where a loop that reads byte by byte and tests after each walks
// five to eight dependent LDS round trips deep into nearly every window (rand4 m = 32: 66 %, rand2: 38 %).
// ---------------------------------------------------------------------------
template <int THREADS, int L, bool LONG, int Q, int GRAM = 0>  // LONG: m > 32, prefix hits are verified; GRAM: bndm_gram above
__global__ __launch_bounds__(THREADS) void bndm_scan(ScanArgs a1, uint64_t tile_first,
                                                     uint32_t ntiles, const BatchItem* __restrict__ batch)
{
    const ScanArgs a = pick_args(a1, batch);  // a pattern set in one grid: blockIdx.y = pattern (launch_batch)
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    if constexpr (GRAM != 0) {
        bndm_gram<THREADS, L, LONG, Q, GRAM>(a, tile_first, ntiles, smem);
        return;
    }
    constexpr int TB = THREADS * L;
    using CT = ColTile<THREADS>;  // a window reaches 31 bytes back: the 32 bytes in front of every segment
    static_assert(L == 64 && (Q == 1 || Q == 2 || Q == 4 || Q == 8), "Q divides 32: no read leaves the window's 32 bytes");
    const uint32_t m = a.m, w = m < 32 ? m : 32;
    uint32_t* B = reinterpret_cast<uint32_t*>(smem);
    constexpr uint32_t kTxt = 1024;
    uint8_t* txt = smem + kTxt;

    // masks left-aligned (B'[c] = B[c] << (32-w)): D << 1 then drops factors that can no longer become a prefix, instead
    // of carrying dead bits above bit w-1 as bndm.c's 32-bit word does for m < 32 (the next AND clears them either way:
    // same D & B, same count) — and "a prefix" is the sign bit
    for (uint32_t i = threadIdx.x; i < 256; i += THREADS)
        B[i] = reinterpret_cast<const uint32_t*>(a.blob + kTableOff)[i] << (32 - w);
    if ((uint32_t)(uintptr_t)(lds_u8_t*)smem != 0u) {  // the walk below addresses LDS by offset
        if (threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(reinterpret_cast<unsigned long long*>(a.count), 1ull << 62);
        return;
    }

    const uint64_t e_begin = a.s_begin + w - 1, e_end = a.s_end + w - 1;
    uint32_t hits = 0;
    uint4 pre[4], ph;  // prefetch registers: 4 tile rows + (threads 0, 1) the 32 bytes in front of the tile
    auto issue = [&](uint64_t tile0) {
        const uint8_t* src = a.text + tile0 + threadIdx.x * 16u;
        pre[0] = ld_stream16(src);
        pre[1] = ld_stream16(src + THREADS * 16);
        pre[2] = ld_stream16(src + THREADS * 32);
        pre[3] = ld_stream16(src + THREADS * 48);
        if (threadIdx.x < 2) ph = ld_stream16(src - 32);
    };
    const uint64_t t_end = tile_first + ntiles;
    uint64_t t = tile_first + blockIdx.x;
    issue(t * TB);
    // the lane's column; a cursor is a byte position in it: 32 + x for byte x of the segment, 0..31 the bytes before
    const uint32_t col4 = kTxt + CT::col(threadIdx.x) * 4u;
    for (; t < t_end; t += gridDim.x) {
        const uint64_t tile0 = t * TB;
        __syncthreads();
        CT::park(txt, pre, ph);
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
        // The lane's walk over its window ends.  HOW says what an occurrence of the w-byte window costs: 0 (m <= 32) it IS
        // an occurrence of P; 1 (LONG, the walk every tile takes) it is counted and its window end remembered — two
        // VALU ops and no branch, where verifying on the spot put a wave-uniform test on every iteration (3 points on the
        // m > 32 cells); 2 (LONG, the lanes that saw more than one in this tile — periodic texts) P[32..m) is compared
        // on the spot (bndm.c:99-102).
        uint32_t nocc = 0, last = 0;
        auto walk = [&](auto how) {
            constexpr int HOW = decltype(how)::value;
            uint32_t e = 32u + x0, k = 0, D = 0xFFFFFFFFu;
            const uint32_t ehi = 32u + x1;
            while (e < ehi) {
                // the window's next Q bytes, T[e-k-Q+1 .. e-k]: byte Q-1 of X is the one bndm.c:50 reads first (right to left)
                uint32_t xw[2] = {0u, 0u};
                {
                    const uint32_t pl = e - k - (Q - 1);  // position of the lowest of them
                    const uint32_t at = col4 + (pl >> 2) * CT::RS;
                    if (Q == 1) {
                        xw[0] = *(const lds_u8_t*)(size_t)(at + (pl & 3u));
                    } else {
                        const uint32_t w0 = *(const lds_u32_t*)(size_t)at;
                        const uint32_t w1 = *(const lds_u32_t*)(size_t)(at + CT::RS);
                        xw[0] = __builtin_amdgcn_alignbyte(w1, w0, pl);
                        if (Q == 8) {
                            const uint32_t w2 = *(const lds_u32_t*)(size_t)(at + 2 * CT::RS);
                            xw[1] = __builtin_amdgcn_alignbyte(w2, w1, pl);
                        }
                    }
                }
                uint32_t G = 0xFFFFFFFFu;
#pragma unroll
                for (int j = 0; j < Q; ++j) {  // step j reads byte Q-1-j; its mask meets D after Q-1-j more shifts
                    const int i = Q - 1 - j;
                    const uint32_t c = (xw[i >> 2] >> (8 * (i & 3))) & 0xFFu;
                    G &= *(const lds_u32_t*)(size_t)(4u * c) << i;  // B[c]
                }
                const uint32_t tt = (D << (Q - 1)) & G;  // bndm.c:51, Q times
                const uint32_t kq = k + Q;
                const bool occ = (int32_t)tt < 0 && kq == w;  // bndm.c:55: all w bytes are read and the factor alive is P[0..w) itself
                D = tt << 1;                                  // bndm.c:57
                const bool done = D == 0;  // no factor alive, or the window is read through
                if (HOW == 0) {
                    hits += occ;
                } else if (HOW == 1) {
                    nocc += occ;
                    last = occ ? e : last;
                } else if (occ) {  // = text + s + w; inside the text because s < s_end
                    hits += global_equal(a.text + seg + (e - 32u) + 1, a.blob + w, m - w);
                }
                // the window ends.  tt == 0: the kq bytes are no factor of P, the kq-1 after their first may be a prefix: move by
                // w - (kq-1) (bndmq4.c:61); tt != 0 — its sign bit alone, or D' would not be 0 — they ARE a prefix of P: move by
                // w - kq (bndm.c:54), by 1 after an occurrence
                e += done ? w - kq + ((tt == 0 || kq == w) ? 1u : 0u) : 0u;
                k = done ? 0u : kq;
                D = done ? 0xFFFFFFFFu : D;
            }
        };
        if (!LONG) {
            walk(std::integral_constant<int, 0>());
        } else {
            walk(std::integral_constant<int, 1>());
            if (__any(nocc != 0)) {  // rare, wave-uniform, once per tile: the 32-byte prefix matched somewhere
                if (nocc > 1) walk(std::integral_constant<int, 2>());
                hits += wave_verify(nocc == 1, a.text + seg + (last - 32u) + 1, a.blob + w, m - w);
            }
        }
    }
    flush_hits(hits, a.count, smem, a.text);
}


// ---------------------------------------------------------------------------
// launcher: BNDM (and BNDML's m <= 32): q bytes of a window per iteration, q = a.halo from the plan (api.cpp build_blob).
// Workgroups per CU (26.6 KB of LDS each, six fit), measured on 1 GiB (ms): a streaming scan (a.sparse,
// rand128 m = 16 / 32 / 256) 0.170 / 0.160 / 0.173 with FOUR (five: 0.176 / 0.178 / 0.184); where windows
// survive — English m = 16 / 32 / 256: 0.179 / 0.179 / 0.185 with FIVE (four: 0.179 / 0.174 / 0.204, six:
// 0.183 / 0.182 / 0.192); a small alphabet (q >= 4), rand4 m = 16 / 32, rand2 m = 32: 0.185 / 0.172 / 0.195
// with SIX (four: 0.202 / 0.165 / 0.214).  Two-wave workgroups (tune(2,2)) were never ahead: 8 of them
// 0.190 / 0.171 / 0.200 on the small alphabets, 0.177 / 0.172 / 0.180 on English.
// ---------------------------------------------------------------------------
hipError_t launch_bndm(const ScanArgs& a, int num_cus, hipStream_t stream, TextCodes codes)
{
    const uint32_t m = a.m, w = m < 32 ? m : 32;
    uint32_t q = (g_tune[1] && g_tune[1] != 9) ? (uint32_t)g_tune[1] : a.halo;  // tune(1, q): experiments (9: the plan's q, no gram table)
    while (q > 1 && w % q) q /= 2;
    const bool two_wave = g_tune[2] == 2;
    const int wgs = a.sparse ? 4 : q >= 4 ? 6 : 5;
    // A text of at most four byte values: bndm_gram (the kernel's comment) — eight one-bit symbols per gram on two values
    // (windows of 8+ bytes), four two-bit symbols on up to four (4+ bytes).  tune(1, 9): never (A/B).
    const int gram = g_tune[1] == 9 ? 0 : ((codes.one & 0xFFu) != 0xFFu && w >= 8) ? 1 : (codes.shift < 7 && w >= 4) ? 2 : 0;
    if (gram) {
        // workgroups per CU, measured on 1 GiB of sigma 2 / 4 (ms; 6 / 5 / 4 / 3 per CU): m = 8: 0.181 / 0.190 / 0.208 / 0.245;
        // m = 16: 0.189 / 0.181 / 0.184 / 0.207; m = 32, 256: 0.182 / 0.180 / 0.167 / 0.187 — a window of 32 bytes moves by ~26: a
        // streaming scan, best with four as the tile kernels on large alphabets are
        const int gwgs = w >= 32 ? 4 : w >= 16 ? 5 : 6;
        const size_t lds = 1056 + ColTile<kBndmT>::bytes();
        const TileRange tr = tiles_for(a.s_begin + w - 1, a.s_end + w - 1, (uint64_t)kBndmT * kBndmL);
        if (gram == 1) {
            if (m > 32) return launch_tiled(bndm_scan<kBndmT, kBndmL, true, 8, 1>, a, tr, kBndmT, lds, gwgs, num_cus, stream);
            return launch_tiled(bndm_scan<kBndmT, kBndmL, false, 8, 1>, a, tr, kBndmT, lds, gwgs, num_cus, stream);
        }
        if (m > 32) return launch_tiled(bndm_scan<kBndmT, kBndmL, true, 4, 2>, a, tr, kBndmT, lds, gwgs, num_cus, stream);
        return launch_tiled(bndm_scan<kBndmT, kBndmL, false, 4, 2>, a, tr, kBndmT, lds, gwgs, num_cus, stream);
    }
#define SG_BNDM(T_, WGS_, Q_)                                                                            \
    do {                                                                                                  \
        const size_t lds = 1024 + ColTile<T_>::bytes();                                                   \
        const TileRange tr = tiles_for(a.s_begin + w - 1, a.s_end + w - 1, (uint64_t)(T_) * kBndmL);      \
        if (m > 32) return launch_tiled(bndm_scan<T_, kBndmL, true, Q_>, a, tr, T_, lds, WGS_, num_cus, stream); \
        return launch_tiled(bndm_scan<T_, kBndmL, false, Q_>, a, tr, T_, lds, WGS_, num_cus, stream);     \
    } while (0)
    if (two_wave) {
        if (q == 8) SG_BNDM(kBndmBusyT, 2 * wgs, 8);
        if (q == 4) SG_BNDM(kBndmBusyT, 2 * wgs, 4);
        if (q == 2) SG_BNDM(kBndmBusyT, 2 * wgs, 2);
        SG_BNDM(kBndmBusyT, 2 * wgs, 1);
    }
    if (q == 8) SG_BNDM(kBndmT, wgs, 8);
    if (q == 4) SG_BNDM(kBndmT, wgs, 4);
    if (q == 2) SG_BNDM(kBndmT, wgs, 2);
    SG_BNDM(kBndmT, wgs, 1);
#undef SG_BNDM
}


}  // namespace sg
