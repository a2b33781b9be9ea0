// gram_skip.hpp — the skip loop of Horspool and Boyer-Moore on GRAMS (hor_scan_gram, k_horg.hip; bm_scan_gram, k_bmg.hip): texts
// of at most four distinct byte values.  One device function, two kernels — each in a translation unit of its own.
#pragma once
#include "dev_common.hpp"
#include "launch_common.hpp"

namespace sg {

// ---------------------------------------------------------------------------
// Horspool  (reference: src/algos/hor.c:26-51) with its bad-character rule taken on the window's last Q BYTES instead of
// its last byte — the super-alphabet form Lecroq's hash3/5/8.c use with a hash; here, on a TEXT of at most four distinct
// byte values (TextCodes: what a text consists of is known since it was created), the gram itself is the index: Q = 8
// one-bit symbols on a text of two values (GRAM = 1), Q = 4 two-bit symbols on three or four (GRAM = 2) — 256 grams.
//   hbc[g] = m - Q - i  for the rightmost i < m - Q with P[i..i+Q) == g   (hor.c:28-29 on grams),
//   hbc[g] = m - Q + 1  for a gram that does not occur there              (hash3.c:44: the default shift),
//   a window is compared (hor.c:41-46) only when its last gram is P's last gram.
// On a byte alphabet of two symbols hor.c's table holds two shifts, a byte or two (hor_scan there: 0.16-0.3 of the
// roofline at any m); the gram table shifts by ~m - Q for nine windows in ten.
// The workgroup builds the table before it starts: its threads walk the pattern's last positions (all of them up to
// kHorGramScan; a longer pattern's earlier occurrences only bound the shift from above, and a smaller shift is safe),
// every position's gram is written with atomicMax into the table's slot — the rightmost wins.
// Tiles as bndm_scan's: column tiles, a lane's 64 window ENDS down its own LDS bank behind the 32 bytes before them;
// ONE gram read + ONE lookup per window.  A candidate's other m - Q bytes are compared in LDS when the window lies in
// the column (m <= 32), else in memory (the first candidate of a tile parked for wave_verify).
// Boyer-Moore (BM = true; reference: src/algos/bm.c:27-93) joins the gram's shift with its good-suffix rule exactly where
// bm.c:89 joins the bad character's: a window whose last gram is not the pattern's differs at the k-th symbol from the
// right, k < Q — a function of the gram, folded into the table: max(hbc[g], bmGs[m-1-k]); a candidate is compared right to
// left (bm.c:83-84) through its 24 nearest bytes in LDS and moves by max(hbc, bmGs) of the byte that differed, by bmGs[0]
// after an occurrence (bm.c:86).
// ---------------------------------------------------------------------------
constexpr uint32_t kHorGramScan = 256;  // pattern positions (from the end) the table is built from: one per thread — a lane owns 64
                                         // window ends, a shift beyond that moves it out of its segment either way (2048 positions and a
                                         // byte-wise check of the whole pattern cost a 4096-byte pattern 0.17 ms per GiB in table building)

template <int THREADS, int L, bool LONG, int Q, int GRAM, bool BM>  // LONG: m > 32 — the window does not lie in the lane's column
__device__ __forceinline__ void gram_skip_scan(const ScanArgs& a, uint64_t tile_first, uint32_t ntiles, uint8_t* smem)
{
    constexpr int TB = THREADS * L;
    using CT = ColTile<THREADS>;
    static_assert(L == 64 && THREADS == 256 && ((GRAM == 1 && Q == 8) || (GRAM == 2 && Q == 4)), "a gram is 8 one-bit or 4 two-bit symbols");
    const uint32_t m = a.m;  // m >= Q (the launchers)
    uint32_t* E = reinterpret_cast<uint32_t*>(smem);
    // LDS: E[256] | column tile | 32 pattern bytes | the flag | BM: u16 gsT[40], gs[0], the parked windows' safe shift (84 bytes)
    // (the tile at a multiple of 256 — its rows are then ds_read2st64's immediate offsets, no add on the lane's address —, the
    // pattern's bytes, the flag and BM's shifts BEHIND it: the LDS of a workgroup is what it was, six of them per CU)
    constexpr uint32_t kTxt = 1024, kPat = kTxt + ColTile<THREADS>::bytes(), kGs = kPat + 48;
    uint8_t* txt = smem + kTxt;
    // the text's codes, from the first words of the text's own allocation (TextCodes, kernels.hpp)
    const uint32_t* const tc = reinterpret_cast<const uint32_t*>(a.text - kFrontPad);
    const uint32_t cshift = GRAM == 2 ? tc[0] : tc[2] & 0xFFu;  // two-bit codes: (c >> shift) & 3; one-bit: (c >> bit) & 1
    const uint32_t rot4 = (cshift + 30u) & 31u;                 // GRAM 2: rotate right by shift - 2 (left by 2 - shift): a byte's code at its bits 2-3
    const uint32_t symtab = GRAM == 2 ? tc[1] : tc[2] >> 8;     // the byte value of each code
    constexpr uint32_t kBits = GRAM == 2 ? 2u : 1u, kMask = GRAM == 2 ? 3u : 1u;
    uint32_t* const foreign = reinterpret_cast<uint32_t*>(smem + kPat + 32);  // set if a pattern byte is no symbol of the text
    E[threadIdx.x] = 0;
    if (threadIdx.x == 0) *foreign = 0;
    // the pattern bytes in front of its last gram that a candidate is compared with in LDS: all m - Q of them while the window
    // lies in the column (m <= 32), the 24 nearest otherwise — what is left of a longer window is compared in memory only when
    // those agree (a candidate in 256 windows is cheap to test in LDS and dear in memory: verifying every one of them there
    // cost a 4096-byte pattern 0.12 ms per GiB)
    const uint32_t nlds = LONG ? 24u : m - Q, lds_from = m - Q - nlds;
    if (threadIdx.x < 32) smem[kPat + threadIdx.x] = threadIdx.x < nlds ? a.blob[lds_from + threadIdx.x] : 0;
    if (BM && threadIdx.x < 42) {  // bm.c:54-66's good-suffix shifts of the pattern's last positions: gsT[k] = bmGs[m-1-k]; [40] = bmGs[0]; [41] = the safe shift
        const uint16_t* gs = reinterpret_cast<const uint16_t*>(a.blob + kTableOff + 1536);
        const uint32_t k = threadIdx.x;
        reinterpret_cast<uint16_t*>(smem + kGs)[k] = k < 40 ? (k < m ? gs[m - 1 - k] : gs[0]) : k == 40 ? gs[0] : gs[m];
    }
    if ((uint32_t)(uintptr_t)(lds_u8_t*)smem != 0u) {  // the walk below addresses LDS by offset
        if (threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(reinterpret_cast<unsigned long long*>(a.count), 1ull << 62);
        return;
    }
    __syncthreads();
    // the gram of the pattern at position i: byte j's code in bits [kBits * j, kBits * (j + 1))
    auto pgram = [&](uint32_t i) -> uint32_t {
        uint32_t g = 0;
#pragma unroll
        for (int j = 0; j < Q; ++j) g |= ((a.blob[i + j] >> cshift) & kMask) << (kBits * j);
        return g;
    };
    {
        // positions i < m - Q, the last kHorGramScan of them: slot[g] = 1 + the rightmost i (0: the gram does not occur)
        const uint32_t npos = m - Q, lo = npos > kHorGramScan ? npos - kHorGramScan : 0u;
        for (uint32_t i = lo + threadIdx.x; i < npos; i += THREADS) atomicMax(&E[pgram(i)], i + 1u);
        // a pattern byte that is no symbol of the text cannot occur in it, and its code is some symbol's: no window is a candidate then
        // (sixteen bytes per thread and step: the pattern slot is 4224 zero-padded bytes; bytes beyond m are not looked at)
        for (uint32_t i0 = threadIdx.x * 16u; i0 < m; i0 += THREADS * 16u) {
            const uint4 v = *reinterpret_cast<const uint4*>(a.blob + i0);
            const uint32_t d[4] = {v.x, v.y, v.z, v.w};
            uint32_t bad = 0;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const uint32_t c = (d[q >> 2] >> (8 * (q & 3))) & 0xFFu;
                bad |= (i0 + q < m && ((symtab >> (8u * ((c >> cshift) & kMask))) & 0xFFu) != c) ? 1u : 0u;
            }
            if (bad) atomicOr(foreign, 1u);
        }
        __syncthreads();
        const uint32_t g = threadIdx.x, at = E[g];
        // (a gram unseen in a long pattern's scanned part may occur before it: the scanned length is a safe shift)
        uint32_t shift = at ? npos - (at - 1u) : lo ? kHorGramScan : npos + 1u;
        const uint32_t last = pgram(npos);
        if (BM && g != last) {
            // bm.c:83-89 right to left: the gram's symbols that agree with the pattern's last ones, then the good-suffix shift of
            // the first that does not — joined with the gram's own shift as bm.c:89 joins it with the bad character's
            uint32_t k = 0;
            while (k < (uint32_t)Q && (((g ^ last) >> (kBits * (Q - 1 - k))) & kMask) == 0) ++k;
            const uint32_t gsk = reinterpret_cast<const uint16_t*>(smem + kGs)[k];
            shift = shift > gsk ? shift : gsk;
        }
        __syncthreads();
        E[g] = ((g == last && *foreign == 0 ? 1u : 0u) << 31) | shift;
    }

    const uint64_t e_begin = a.s_begin + m - 1, e_end = a.s_end + m - 1;
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
    auto text4 = [&](uint32_t p) -> uint32_t {  // four text bytes from position p of the lane's column
        const uint32_t at = col4 + (p >> 2) * CT::RS;
        return __builtin_amdgcn_alignbyte(*(const lds_u32_t*)(size_t)(at + CT::RS), *(const lds_u32_t*)(size_t)at, p);
    };
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
        // column and shifts the two dwords apart, and the loop is bound by what it issues (no e - 3, no + 32 for the tile's
        // offset, the codes scaled by 4 as they are extracted: the dot product is the table's byte offset).
        uint32_t pl = 32u + x0 - (Q - 1);
        const uint32_t plhi = 32u + x1 - (Q - 1);
        uint32_t parked_e = 0;  // LONG: the window end of the tile's first candidate (0: none; e >= 32)
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
            uint32_t cand_shift = 1;  // BM: what a candidate moves by
            (void)cand_shift;
            {   // the window's last gram is P's: hor.c:41-46 for those lanes (a wave without one skips the block)
                if ((int32_t)ent < 0) {
                    const uint32_t e = pl + (Q - 1);
                    const uint32_t ws = e - (Q - 1) - nlds;
                    if (!BM) {
                        // the nlds bytes in front of the gram against the pattern's, a dword at a time
                        bool ok = true;
                        for (uint32_t d = 0; d < nlds; d += 4) {
                            const uint32_t nb = nlds - d < 4 ? nlds - d : 4u;
                            const uint32_t mask = nb == 4 ? 0xFFFFFFFFu : (1u << (8u * nb)) - 1u;
                            ok = ok && ((text4(ws + d) ^ *(const lds_u32_t*)(size_t)(kPat + d)) & mask) == 0;
                        }
                        if (ok) {
                            if (!LONG) ++hits;
                            else if (parked_e == 0) parked_e = e;
                            else hits += global_equal(a.text + seg + (e - 32u) - (m - 1), a.blob, lds_from);  // = text + s: the window's first bytes
                        }
                    } else {
                        // Boyer-Moore's order (bm.c:83-84): right to left, the dword nearest the gram first, until one differs; k = bytes
                        // of the window that agreed, its last gram included
                        uint32_t k = Q;
                        bool ok = true;
                        for (int32_t d = (int32_t)((nlds - 1u) & ~3u); d >= 0 && ok; d -= 4) {
                            const uint32_t nb = nlds - (uint32_t)d < 4 ? nlds - (uint32_t)d : 4u;
                            const uint32_t mask = nb == 4 ? 0xFFFFFFFFu : (1u << (8u * nb)) - 1u;
                            const uint32_t x = (text4(ws + (uint32_t)d) ^ *(const lds_u32_t*)(size_t)(kPat + (uint32_t)d)) & mask;
                            if (x) {
                                k += nb - 1u - ((31u - (uint32_t)__builtin_clz(x)) >> 3);  // the bytes above the highest one that differs
                                ok = false;
                            } else {
                                k += nb;
                            }
                        }
                        const uint16_t* gsT = reinterpret_cast<const uint16_t*>(smem + kGs);
                        uint32_t sh = ent & 0x7FFFFFFFu;  // the gram's shift (hor.c:49 on grams), joined with ...
                        if (!ok) {
                            const uint32_t gsk = gsT[k];  // ... bm.c:89's good-suffix shift of the position that differed
                            sh = sh > gsk ? sh : gsk;
                        } else if (!LONG) {
                            ++hits;
                            sh = gsT[40];                 // bm.c:86: after an occurrence
                        } else {
                            if (parked_e == 0) parked_e = e;
                            else hits += global_equal(a.text + seg + (e - 32u) - (m - 1), a.blob, lds_from);
                            sh = gsT[41];                 // a shift that is safe whatever the rest of the window says (api.cpp build_blob)
                        }
                        cand_shift = sh ? sh : 1u;
                    }
                }
            }
            pl += (BM && (int32_t)ent < 0) ? cand_shift : ent & 0x7FFFFFFFu;  // hor.c:49 on grams
        }
        if (LONG && __any(parked_e != 0)) hits += wave_verify(parked_e != 0, a.text + seg + (parked_e - 32u) - (m - 1), a.blob, lds_from);
    }
    flush_hits(hits, a.count, smem, a.text);
}


}  // namespace sg
