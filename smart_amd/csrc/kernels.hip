// kernels.hip — gfx950 (CDNA4, wave64) scan kernels for exact string matching.
//
// Every kernel computes the same function
//     count(P,T) = |{ s in [s_begin, s_end) : T[s..s+m) == P }|
// (overlapping occurrences count) and differs in the per-lane scan strategy and in
// the tables it stages in LDS.  Three families:
//
//  1. LDS TILES — the skip algorithms HOR, BM, BNDM (hor_scan, bm_scan, bndm_scan).
//     The text is cut into tiles of TB = THREADS*L bytes on ABSOLUTE text offsets, so
//     every tile load is 16-byte aligned and coalesced (global_load_dwordx4 nt, 1 KiB
//     per wave-load); tile t+1 is prefetched into registers while the lanes walk tile
//     t.  Tiles are indexed by window END position e = s+m-1 with a BACK halo of
//     H = min(m-1, 16) bytes: the byte that drives the shift, T[e], is always in the
//     tile; a lane verifies right-to-left through the halo by itself and — only for
//     m-1 > H and only after H+1 bytes matched — parks the window for a
//     wave-cooperative comparison of the rest (wave_verify).
//  2. RUNS THROUGH LDS SLABS — the serial automata SO and KMP (so_runs, kmp_runs).
//     A lane owns a run of 2-4 KiB of start positions (128+ bytes on small texts); the wave
//     fetches the next 128-byte line of each of its 64 runs with coalesced non-temporal loads,
//     parks it in its own LDS slab one 64-byte half at a time and every lane reads its run
//     back.  No workgroup barrier after the table set-up.
//  3. PACKED — EPSM, and the short-pattern / tiny-shift regime of the skip algorithms
//     (packed_scan): every alignment is compared from registers, no LDS.
//
// Restarting an algorithm at a lane / tile / run / GPU boundary preserves the count
// (SURVEY.md §7 restart table): skip algorithms carry no state between windows, the
// automata restart in their initial state and re-scan w-1 bytes.  Per-lane hit
// counters are summed across the 64-lane wave and the workgroup; one 64-bit atomic per workgroup.
// A pattern set over a small text runs as ONE grid, gridDim.y = pattern (launch_scan_set).
// Earlier designs of the serial kernels (so_scan, kmp_scan, so_runs64, so_runs1, kmp_links_runs,
// kmp_runs1) live in kernels_ab.inc and only in the A/B build (make AB=1, smartgpu_tune).
#include "kernels.hpp"

#include <cstdio>
#include <cstdlib>
#include <map>
#include <utility>

#include "../../include/smartgpu.h"

namespace sg {

// ---------------------------------------------------------------------------
// helpers
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t round16(uint32_t x) { return (x + 15u) & ~15u; }

// LDS by OFFSET: a kernel without static LDS has its dynamic segment at offset 0, and an integer offset as the address
// saves the "+ base" the compiler otherwise adds to every LDS address it forms from the extern array (a v_add with a
// relocated 0); the kernels that do this check the base once and poison the count if it is not 0.
typedef __attribute__((address_space(3))) uint32_t lds_u32_t;
typedef __attribute__((address_space(3))) uint16_t lds_u16_t;
typedef __attribute__((address_space(3))) uint8_t lds_u8_t;


// The arguments of this workgroup: the by-value set of a single launch, or — a pattern set in one grid
// (launch_scan_set) — that set with the per-pattern fields of element blockIdx.y of the set's item array: where
// the pattern's tables sit in the arena, which count slot is its own, and what its plan decided.  The pointers
// stay derived from the kernel arguments (base + offset), so the compiler keeps treating them as global memory
// and the values as scalars.  (Selecting between a by-value ScanArgs and one loaded from memory made every
// load of the kernels a flat_load and moved their address arithmetic to the vector unit: packed_scan 12-15 % slower.)
__device__ __forceinline__ ScanArgs pick_args(const ScanArgs& a1, const BatchItem* __restrict__ batch)
{
    ScanArgs a = a1;
    if (batch) {
        constexpr int W = sizeof(BatchItem) / 4;
        static_assert(sizeof(BatchItem) == 32, "BatchItem is copied word by word");
        uint32_t w[W];
        __builtin_memcpy(w, batch + blockIdx.y, sizeof(BatchItem));
#pragma unroll
        for (int i = 0; i < W; ++i) w[i] = __builtin_amdgcn_readfirstlane(w[i]);
        BatchItem it;
        __builtin_memcpy(&it, w, sizeof(BatchItem));
        a.blob = a1.blob + it.blob_off;
        a.count = a1.count + it.count_idx;
        a.halo = it.halo;
        a.fp_off = it.fp_off;
        a.prefer_packed = it.prefer_packed;
        a.sparse = it.sparse;
        a.so_off = it.so_off;
    }
    return a;
}

// 16-byte load of text that is read once: non-temporal (global_load_dwordx4 ... nt).
// Measured with tools/probe/read_bw.hip on MI355X: a coalesced streaming read reaches
// 7.0-7.1 TB/s with nt loads against 6.2-6.3 TB/s with the default cache policy.
__device__ __forceinline__ uint4 ld_stream16(const uint8_t* p)
{
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 v;
    v.x = __builtin_nontemporal_load(&q->x);
    v.y = __builtin_nontemporal_load(&q->y);
    v.z = __builtin_nontemporal_load(&q->z);
    v.w = __builtin_nontemporal_load(&q->w);
    return v;
}

// Stage nbytes (multiple of 16) from 16-byte-aligned global memory to
// 16-byte-aligned LDS, 16 B per lane per step (coalesced 1 KiB per wave-load).
template <int THREADS>
__device__ __forceinline__ void stage_bytes(uint8_t* __restrict__ lds,
                                            const uint8_t* __restrict__ src, uint32_t nbytes)
{
    for (uint32_t o = threadIdx.x * 16u; o < nbytes; o += THREADS * 16u)
        *reinterpret_cast<uint4*>(lds + o) = ld_stream16(src + o);
}

// Fixed-size variant: TB bytes with all loads issued before the LDS stores.
template <int THREADS, int TB>
__device__ __forceinline__ void stage_tile(uint8_t* __restrict__ lds,
                                           const uint8_t* __restrict__ src)
{
    constexpr int N = TB / (THREADS * 16);
    static_assert(TB % (THREADS * 16) == 0, "tile must be whole 16-byte rows");
    uint4 v[N];
#pragma unroll
    for (int k = 0; k < N; ++k)
        v[k] = ld_stream16(src + (k * THREADS + threadIdx.x) * 16);
#pragma unroll
    for (int k = 0; k < N; ++k)
        *reinterpret_cast<uint4*>(lds + (k * THREADS + threadIdx.x) * 16) = v[k];
}

// Sum the per-lane hit counters over the workgroup; ONE atomic per workgroup.  (One per wave was
// the first version: on dense hits — short patterns, small alphabets — thousands of atomics on the
// same result slot serialise behind each other at the end of the kernel.)  `lds` is any 8-byte
// aligned 128 bytes of the kernel's LDS: every wave is past its last use of the LDS when it gets
// here, which the first barrier establishes for the whole workgroup.
__device__ __forceinline__ void flush_hits(uint32_t lane_hits, unsigned long long* out, void* lds)
{
    unsigned long long v = lane_hits;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    unsigned long long* part = static_cast<unsigned long long*>(lds);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long sum = 0;
        for (uint32_t w = 0; w < blockDim.x / 64; ++w) sum += part[w];
        if (sum != 0) atomicAdd(out, sum);
    }
}

// 16 bytes at a (text, any alignment) vs 16 bytes at b (pattern slot), first `nb`
// bytes only (1 <= nb <= 16).  Both reads stay in bounds by construction: the text
// buffer has a back pad and the pattern slot of the blob is 4224 zero-padded bytes.
// (gfx950 global loads may be unaligned: an align-1 16-byte copy compiles to one
// global_load_dwordx4.)
__device__ __forceinline__ bool differ16(const uint8_t* __restrict__ a,
                                         const uint8_t* __restrict__ b, uint32_t nb)
{
    uint4 x, y;
    __builtin_memcpy(&x, a, 16);
    __builtin_memcpy(&y, b, 16);
    const uint32_t d[4] = {x.x ^ y.x, x.y ^ y.y, x.z ^ y.z, x.w ^ y.w};
    uint32_t acc = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int left = (int)nb - 4 * q;  // bytes of dword q that count
        const uint32_t mask = left >= 4 ? 0xFFFFFFFFu : left <= 0 ? 0u : (0xFFFFFFFFu >> (8 * (4 - left)));
        acc |= d[q] & mask;
    }
    return acc != 0;
}

// Equality of `len` bytes in memory, 16 bytes per step.
__device__ __forceinline__ bool global_equal(const uint8_t* __restrict__ a,
                                             const uint8_t* __restrict__ b, uint32_t len)
{
    for (uint32_t i = 0; i < len; i += 16)
        if (differ16(a + i, b + i, len - i < 16 ? len - i : 16u)) return false;
    return true;
}

// Wave-cooperative verification.  A candidate that survived the in-LDS filter of
// a long pattern still needs `len` more bytes compared in memory; done by its own
// lane that is a serial chain of dependent loads (0.5 ms for ONE m=4096 match —
// measured), so each lane parks its first candidate of a tile and, at a
// wave-uniform point, the 64 lanes compare 1 KiB per step together.
// Returns 1 in the lane whose candidate verified, 0 elsewhere.
__device__ __forceinline__ uint32_t wave_verify(bool has, const uint8_t* tptr,
                                                const uint8_t* __restrict__ pptr, uint32_t len)
{
    unsigned long long todo = __ballot(has);
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t mine = 0;
    while (todo) {
        const int src = __builtin_ctzll(todo);  // wave-uniform
        todo &= todo - 1;
        const unsigned long long tp = (unsigned long long)tptr;
        const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)tp, src);
        const uint32_t hi = __builtin_amdgcn_readlane((uint32_t)(tp >> 32), src);
        const uint8_t* t = reinterpret_cast<const uint8_t*>(((unsigned long long)hi << 32) | lo);
        bool diff = false;
        for (uint32_t off = lane * 16u; off < len; off += 1024u)
            diff |= differ16(t + off, pptr + off, len - off < 16 ? len - off : 16u);
        if (!__any(diff) && lane == (uint32_t)src) mine = 1;
    }
    return mine;
}

// Flat tiles are stored dword-swizzled: byte i of the tile region sits at i ^ ((i >> 5) & 0x3C), i.e. the
// dword index inside its 64-byte block is XORed with bits 7..10 of i.  A lane owns 64 consecutive bytes,
// so at the same offset x the 64 lanes' addresses differ by multiples of 64 bytes = 16 dwords: unswizzled
// that is TWO banks for the whole wave — and on random text over a large alphabet the lanes do move in
// lockstep (nearly every shift is m).  Swizzled, 32 consecutive lanes cover the 32 banks.
static __device__ __forceinline__ uint32_t tile_at(uint32_t i) { return i ^ ((i >> 5) & 0x3Cu); }

// park one 16-byte chunk (unswizzled byte index i0, a multiple of 16) of the tile region
static __device__ __forceinline__ void tile_park(uint8_t* txt, uint32_t i0, const uint4& v)
{
    uint32_t* blk = reinterpret_cast<uint32_t*>(txt + (i0 & ~63u));
    const uint32_t d = (i0 >> 2) & 15u, s = (i0 >> 7) & 15u;
    blk[(d + 0) ^ s] = v.x;
    blk[(d + 1) ^ s] = v.y;
    blk[(d + 2) ^ s] = v.z;
    blk[(d + 3) ^ s] = v.w;
}

// Lane tiles (bm_scan, hor_flat): the 64 text bytes a lane owns sit CONTIGUOUSLY in LDS behind a private copy of
// the DUP = HALO - 4 bytes before them, HALO + 64 bytes per lane — an odd number of dwords, so lanes at equal
// offsets (a streaming scan on a large alphabet moves them in lockstep) cover all 32 banks without a swizzle.
//   * the address of T[e - k] is one subtraction from the lane's cursor (tile_at: four VALU ops per read);
//   * bytes are contiguous (a q-gram could be ONE unaligned ds_read_b32 / _b64 — bndm_scan tried: the LDS stalls on them,
//     see ColTile);
//   * the price: the last DUP bytes of every segment are parked twice (16 more ds_write_b32 per tile in a quarter or
//     half of the lanes) and a tile takes (64 + HALO) / 64 of its size in LDS.
// Byte x (0..63) of segment s is at s * STRIDE + HALO + x; bytes [4, HALO) of a segment's region are T[seg - DUP, seg),
// bytes [0, 4) are padding (never filled; whoever reads them ignores what they hold).
template <int HALO>
struct LaneTile {
    static_assert(HALO % 16 == 4 && ((64 + HALO) / 4) % 2 == 1, "16 or 32 duplicated bytes + 4 of padding, odd dword stride");
    static constexpr uint32_t STRIDE = 64 + HALO, DUP = HALO - 4;
    static __host__ __device__ constexpr uint32_t bytes(uint32_t segments) { return segments * STRIDE; }
    // park the j-th 16-byte chunk of the tile (segment j / 4, quarter j % 4); the last DUP / 16 quarters of a
    // segment also go in front of the next one
    static __device__ __forceinline__ void park(uint8_t* txt, uint32_t j, const uint4& v, uint32_t segments)
    {
        const uint32_t s = j >> 2, part = j & 3u;
        uint32_t* d = reinterpret_cast<uint32_t*>(txt + s * STRIDE + HALO + 16u * part);
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        if (part >= 4u - DUP / 16u && s + 1 < segments) {
            uint32_t* h = reinterpret_cast<uint32_t*>(txt + (s + 1) * STRIDE + HALO + 16u * part - 64u);
            h[0] = v.x; h[1] = v.y; h[2] = v.z; h[3] = v.w;
        }
    }
    // chunk t (< DUP / 16) of the DUP bytes in front of the tile: segment 0's copy
    static __device__ __forceinline__ void park_front(uint8_t* txt, uint32_t t, const uint4& v)
    {
        uint32_t* h = reinterpret_cast<uint32_t*>(txt + 4u + 16u * t);
        h[0] = v.x; h[1] = v.y; h[2] = v.z; h[3] = v.w;
    }
};

// Column tiles (bndm_scan): the text of a tile as a dword matrix [kColRows rows][THREADS columns], every lane's 64-byte
// segment DOWN its own column behind a copy of the 32 bytes before it — rows 0..7: T[seg-32, seg), rows 8..23: the
// segment, row 24: never filled (a three-dword read may touch it).  A row is THREADS * 4 bytes, a multiple of 128: the
// LDS bank of a dword is its column mod 32 whatever the row, so a wave whose lanes read ANY rows of their own columns
// reads conflict-free (a flat tile, swizzled or padded, serves such a gather in three to four passes, and an unaligned
// ds_read_b32 stalls on top: bndm_scan on lane tiles, rand4 m = 32: LDS 82 % busy, 52 % of that SQ_LDS_UNALIGNED_STALL,
// profiles/r03/c_pmc_bndm_rand4_m32.txt).  A q-gram is two or three ALIGNED dwords a row apart — one ds_read2st64_b32 —
// and v_alignbyte_b32.
// Parking without a transpose in registers: the coalesced loads leave lane (Q, p) = (tid / 4, tid % 4) with quarter p of
// the four segments r * G + Q (r = 0..3, G = THREADS / 4).  Written straight, the four lanes of a quad would hit ONE
// column, one bank, four times; so segment s lives in column col(s) = s with its low five bits rotated by 8 * (s / G),
// and in step t lane (Q, p) writes its quarter of segment ((p + t) % 4) * G + Q: the quads of a half-wave then cover the
// 32 banks exactly once.  Which register that is depends on p: the four chunks are rotated by p once (two conditional
// stages, 32 v_cndmask), after which step t writes register t.
constexpr uint32_t kColRows = 25;
template <int THREADS>
struct ColTile {
    static constexpr uint32_t RS = THREADS * 4u, G = THREADS / 4u;
    static_assert(RS % 128 == 0, "the bank of a dword must not depend on its row");
    static __host__ __device__ constexpr uint32_t bytes() { return kColRows * RS; }
    static __device__ __forceinline__ uint32_t col(uint32_t s) { return (s & ~31u) | ((s + 8u * (s / G)) & 31u); }
    // the lane's four chunks (row r of the tile's coalesced loads in e[r]) and, from threads 0 and 1, the 32 bytes
    // in front of the tile (front)
    static __device__ __forceinline__ void park(uint8_t* txt, uint4 (&e)[4], const uint4& front)
    {
        const uint32_t tid = threadIdx.x, Q = tid >> 2, p = tid & 3u;
        {   // e[t] <- e[(t + p) % 4]
            const bool b0 = p & 1u, b1 = p & 2u;
            const uint4 a0 = e[0], a1 = e[1], a2 = e[2], a3 = e[3];
#define SG_SEL(c_, x_, y_) make_uint4((c_) ? (x_).x : (y_).x, (c_) ? (x_).y : (y_).y, (c_) ? (x_).z : (y_).z, (c_) ? (x_).w : (y_).w)
            const uint4 c0 = SG_SEL(b0, a1, a0), c1 = SG_SEL(b0, a2, a1), c2 = SG_SEL(b0, a3, a2), c3 = SG_SEL(b0, a0, a3);
            e[0] = SG_SEL(b1, c2, c0);
            e[1] = SG_SEL(b1, c3, c1);
            e[2] = SG_SEL(b1, c0, c2);
            e[3] = SG_SEL(b1, c1, c3);
#undef SG_SEL
        }
#pragma unroll
        for (uint32_t t = 0; t < 4; ++t) {
            const uint32_t s = ((p + t) & 3u) * G + Q;  // the segment whose quarter p is in e[t]
            uint32_t* d = reinterpret_cast<uint32_t*>(txt + (8u + 4u * p) * RS + col(s) * 4u);
            d[0] = e[t].x; d[RS / 4] = e[t].y; d[2 * RS / 4] = e[t].z; d[3 * RS / 4] = e[t].w;
            if (p >= 2 && s + 1 < (uint32_t)THREADS) {  // the segment's last 32 bytes: also in front of the next one
                uint32_t* h = reinterpret_cast<uint32_t*>(txt + (4u * (p - 2)) * RS + col(s + 1) * 4u);
                h[0] = e[t].x; h[RS / 4] = e[t].y; h[2 * RS / 4] = e[t].z; h[3 * RS / 4] = e[t].w;
            }
        }
        if (tid < 2) {  // segment 0's copy of T[tile0 - 32, tile0): column col(0) = 0, rows 4 tid ..
            uint32_t* h = reinterpret_cast<uint32_t*>(txt + (4u * tid) * RS);
            h[0] = front.x; h[RS / 4] = front.y; h[2 * RS / 4] = front.z; h[3 * RS / 4] = front.w;
        }
    }
};

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
    flush_hits(hits, a.count, smem);
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
    flush_hits(hits, a.count, smem);
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
    flush_hits(hits, a.count, smem);
}

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
    flush_hits(hits, a.count, smem);
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
template <int THREADS, int L, bool LONG, int Q>  // LONG: m > 32, prefix hits are verified
__global__ __launch_bounds__(THREADS) void bndm_scan(ScanArgs a1, uint64_t tile_first,
                                                     uint32_t ntiles, const BatchItem* __restrict__ batch)
{
    const ScanArgs a = pick_args(a1, batch);  // a pattern set in one grid: blockIdx.y = pattern (launch_batch)
    constexpr int TB = THREADS * L;
    using CT = ColTile<THREADS>;  // a window reaches 31 bytes back: the 32 bytes in front of every segment
    static_assert(L == 64 && (Q == 1 || Q == 2 || Q == 4 || Q == 8), "Q divides 32: no read leaves the window's 32 bytes");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
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
    flush_hits(hits, a.count, smem);
}

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
    flush_hits(hits, a.count, smem);
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
    flush_hits(hits, a.count, smem);
}


// delta[st][c]: one v_perm_b32 builds st*256 + c; the row of state st is stored XOR-swizzled,
// delta[st][c] at st*256 + (c ^ st), which costs one v_xor: every row starts on LDS bank 0, so on
// a small alphabet (few distinct c) lanes in different states would all meet on the same few
// banks (rand2: 33 % -> 42-48 % of 8 TB/s with the swizzle, rand128 unchanged).
__device__ __forceinline__ uint32_t kmp_delta(uint32_t dword, uint32_t st, int byte)
{
    const uint32_t addr = __builtin_amdgcn_perm(dword, st, 0x0c0c0004u + byte) ^ st;
    return *(const lds_u8_t*)(size_t)addr;
}

// ---------------------------------------------------------------------------
// Runs through LDS: a wave FETCHES whole 128-byte cache lines of its 64 runs and PARKS them in
// its slab one 64-byte half at a time.  The first version fetched 64 bytes per run and step;
// with every run of the text in flight at once the two halves of a line were fetched a step
// apart and BOTH missed L2 (PMC: FETCH_SIZE 1.73x the text, TCC_MISS x 128 B = 1.86 GB for
// 1 GiB) — the data path alone, without any automaton work, took 0.236 ms per GiB.  Parking
// whole lines instead (8 KB of slab per wave) fixed the traffic but cost waves (12 per CU
// next to a 64 KB table), and these kernels live on occupancy: one dependent LDS lookup per
// byte.  So the two halves of a line are requested back to back (load i: bytes 0..63 of runs
// 16i + lane/4, load 4+i: bytes 64..127 of the same runs), held in registers, and the slab
// stays [64 runs][64 B]: 4 KB per wave, 16 waves per CU next to any table.
// Tried and dropped (session u): 8 or 12 waves per CU instead of 16: 40-54 %; runs of 256..4096 bytes: within 3
// points of each other.  (Whole-line loads of 8 runs per instruction were equal-or-worse WITHOUT non-temporal
// loads; with them they are so_runs1' loader now — LineIo below.)
// Tried and dropped (measured on rand128, 1 GiB): two runs per lane with interleaved lookups
// (57-59 % against 65-67 %), groups handed out by a device-wide atomic counter (same-address
// atomics serialise at ~16 ns and, returning through vmcnt, stall every wave's first fetch:
// 40-54 %), groups drawn from a per-workgroup LDS counter (no gain over equal static shares).
// Slab layout, unpadded: the 16-byte piece c of run R sits in slot 4R + (c ^ ((R >> 2) & 3)),
// which makes every one of ds_read_b128's 16-lane groups cover 16 distinct slots of the
// 256-byte bank row (no padding, no conflicts).
// ---------------------------------------------------------------------------
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

// Which group of runs a wave starts with (it then strides by the number of waves in the grid).  A text that
// gives every wave of the grid a group: the waves of a workgroup take ADJACENT groups — one contiguous stretch
// of the text per CU (measured: 5-10 % faster than groups a grid apart, whose pages miss the CU's TLB).  Fewer
// groups than waves (a small text, or one whose run length was rounded up): workgroup b takes the groups
// [b*G/B, (b+1)*G/B) — still one contiguous stretch per CU, every CU within one group of the others, and a small
// text's few groups on different CUs (0.97 GiB, 3974 groups for 4096 waves: KMP 60 % -> with this 69 %).
__device__ __forceinline__ uint64_t first_group(uint64_t nruns, uint32_t per_group, uint32_t waves, uint32_t wave)
{
    const uint64_t ngroups = (nruns + per_group - 1) / per_group;
    if (ngroups >= (uint64_t)gridDim.x * waves) return (uint64_t)blockIdx.x * waves + wave;
    const uint64_t lo = blockIdx.x * ngroups / gridDim.x, hi = (blockIdx.x + 1ull) * ngroups / gridDim.x;
    return lo + wave < hi ? lo + wave : ngroups;  // ngroups: none (the caller's loop ends at once)
}

struct RunIo {
    uint8_t* wr;         // where this lane parks its piece of load i (+ 1024*i)
    const uint8_t* rd;   // this lane's own run in the slab
    uint32_t rswz;       // XOR applied to the piece offset 16*c when reading
    uint32_t loff;       // loader role: byte offset of this lane's piece inside a 16-run block
};

__device__ __forceinline__ RunIo run_io(uint8_t* slab, uint32_t lane, uint32_t run_len)
{
    RunIo io;
    // loader role: piece lane&3 of run R = 16i + lane/4, so (R >> 2) & 3 = (lane >> 4) & 3
    io.wr = slab + ((lane >> 2) * 4u + ((lane & 3u) ^ ((lane >> 4) & 3u))) * 16u;
    io.rd = slab + 64u * lane;
    io.rswz = 16u * ((lane >> 2) & 3u);
    io.loff = (lane >> 2) * run_len + 16u * (lane & 3u);
    return io;
}

__device__ __forceinline__ uint4 run_piece(const RunIo& io, int c)
{
    return *reinterpret_cast<const uint4*>(io.rd + ((16u * c) ^ io.rswz));
}

// parking: the four registers of one half into the slab
#define RUN_PARK(io_, r0_, r1_, r2_, r3_)                                          \
    do {                                                                           \
        *reinterpret_cast<uint4*>((io_).wr) = r0_;                                 \
        *reinterpret_cast<uint4*>((io_).wr + 1024) = r1_;                          \
        *reinterpret_cast<uint4*>((io_).wr + 2048) = r2_;                          \
        *reinterpret_cast<uint4*>((io_).wr + 3072) = r3_;                          \
    } while (0)

#define LINE_FETCH_R(gbase_, blk_, off_, r0_, r1_, r2_, r3_, r4_, r5_, r6_, r7_)   \
    do {                                                                           \
        const uint8_t* p_ = (gbase_) + (off_);                                     \
        r0_ = ld_stream16(p_ + (blk_)[0]);                                         \
        r1_ = ld_stream16(p_ + (blk_)[1]);                                         \
        r2_ = ld_stream16(p_ + (blk_)[2]);                                         \
        r3_ = ld_stream16(p_ + (blk_)[3]);                                         \
        r4_ = ld_stream16(p_ + (blk_)[4]);                                         \
        r5_ = ld_stream16(p_ + (blk_)[5]);                                         \
        r6_ = ld_stream16(p_ + (blk_)[6]);                                         \
        r7_ = ld_stream16(p_ + (blk_)[7]);                                         \
    } while (0)
#define LINE_FETCH(gbase_, blk_, off_) LINE_FETCH_R(gbase_, blk_, off_, n0, n1, n2, n3, n4, n5, n6, n7)
// ---- the whole-line loader with half-swapped registers (so_runs, kmp_runs) ----------------------
// LineIo's loads (every 128-byte line requested by ONE non-temporal instruction) with RunIo's parking
// cost.  Load i fetches the lines of runs 8i .. 8i+7 with lane = 32*half + 4*(run in block) + piece:
// lanes 0-31 hold the pieces of the lines' first 64 bytes, lanes 32-63 those of their second.  One
// v_permlane32_swap per dword then exchanges the upper lanes of load 2j with the lower lanes of load
// 2j+1: register 2j now holds FIRST halves in all 64 lanes (runs 16j .. 16j+15, lane = 4*run + piece —
// RunIo's layout), register 2j+1 the second halves.  A half is parked with four full-wave
// ds_write_b128 instead of LineIo's eight half-empty ones: a wave64 ds_write_b128 occupies the LDS
// data path for 13 cycles whatever its EXEC mask (MI355X_MICROARCH.md §LDS), so parking cost
// 1.6 LDS-path cycles per text byte and wave next to 2.0 for the gathers; now 0.8, for 16 swaps per line.
__device__ __forceinline__ RunIo swap_io(uint8_t* slab, uint32_t lane, uint32_t run_len)
{
    RunIo io = run_io(slab, lane, run_len);  // parking and reading are RunIo's
    io.loff = ((lane >> 2) & 7u) * run_len + (lane >> 5) * 64u + 16u * (lane & 3u);
    return io;
}

__device__ __forceinline__ void swap_halves(uint4& lo, uint4& hi)
{
    // v_permlane32_swap vdst, src: lanes 32-63 of vdst <-> lanes 0-31 of src
#define SG_SWAP(f_)                                                                  \
    do {                                                                             \
        const auto r_ = __builtin_amdgcn_permlane32_swap(lo.f_, hi.f_, false, false); \
        lo.f_ = r_[0];                                                               \
        hi.f_ = r_[1];                                                               \
    } while (0)
    SG_SWAP(x);
    SG_SWAP(y);
    SG_SWAP(z);
    SG_SWAP(w);
#undef SG_SWAP
}

#define SWAP_LINE_R(r0_, r1_, r2_, r3_, r4_, r5_, r6_, r7_) \
    do {                           \
        swap_halves(r0_, r1_);     \
        swap_halves(r2_, r3_);     \
        swap_halves(r4_, r5_);     \
        swap_halves(r6_, r7_);     \
    } while (0)
#define SWAP_LINE() SWAP_LINE_R(n0, n1, n2, n3, n4, n5, n6, n7)

// Shift-Or over per-lane runs, FOUR text bytes per step of the recurrence.
//
// so.c:55 is D = (D << 1) | S[c] once per byte.  Four of them are
//     D = (D << 4) | (S[c0] << 3) | (S[c1] << 2) | (S[c2] << 1) | S[c3]
// and the part after the first OR does not depend on D: three v_lshl_or_b32 off the chain, one on it —
// still one VALU op per byte for the recurrence, but the hit test (so.c:56) comes for all four bytes at
// once.  The state is held with the mask's top bit at bit 28 (S'[c] = S[c] << (29 - w), bits 29..31
// zero): after a step, bits 28..31 of D are bit w-1 of the four intermediate states, oldest on top
// (a set bit of S'[c3] cannot reach bit 29, of S'[c2] << 1 not bit 30, ...), and ONE v_alignbit_b32 per
// four bytes moves them into the hit collector.  Three bits of headroom make w = min(m, 29): patterns
// of 30+ bytes are filtered by their 29-byte prefix and verified (so.c:69-96 does that from 33 bytes
// on with a 32-byte prefix; the count is the same).
// Per text byte: one v_perm_b32 (gather address), one ds_read_b32 (bank-private gather), one
// v_lshl_or_b32, a quarter v_alignbit_b32 — 2.25 VALU + 1 LDS against 3 + 1 with a step per byte.
// The loader is swap_io above.  Shift-And (sa.c) counts in complemented form on the same kernel
// (api.cpp build_blob).
// (a << K) | b as ONE v_lshl_or_b32: left to itself the compiler reassociates the OR tree, shifts every
// mask on its own and joins them with v_or3_b32 — 6 VALU ops per four bytes instead of 4.  The empty
// asm hides the value from the reassociation and emits nothing.
template <int K>
__device__ __forceinline__ uint32_t lshl_or_now(uint32_t a, uint32_t b)
{
    uint32_t r = (a << K) | b;
    asm("" : "+v"(r));
    return r;
}

// so_runs, LONG: the hits of one 16-byte chunk (bit 15-q of hm: the w-byte prefix ends at byte base + q of the
// run).  The first one of a half is parked for wave_verify (its offset in the run; 0 = none), further ones —
// rare — are completed by the lane itself.  Out of line: inlined (four times) its loads and loops cost the
// streaming path of the long-pattern instantiation 20+ VGPRs and spills.
// Returns (hits counted << 32) | parked offset: by value, so that the caller's copy stays in a register.
static __device__ __attribute__((noinline)) uint64_t so_long_hits(const uint8_t* run_text, const uint8_t* tail, uint32_t len,
                                                                  uint32_t base, uint32_t hm, uint32_t parked_off)
{
    uint32_t n = 0;
    while (hm) {
        const uint32_t bit = 31u - __builtin_clz(hm);
        hm &= ~(1u << bit);
        const uint32_t off = base + (15u - bit) + 1;  // the byte after the prefix = start + w
        if (parked_off == 0) parked_off = off;
        else n += global_equal(run_text + off, tail, len);
    }
    return ((uint64_t)n << 32) | parked_off;
}

// FOUR — a text of at most four distinct byte values (ScanArgs.four_shift, four_symtab: what the text consists of is
// known since it was created): the per-lane table holds, instead of the 256 masks, the 256 values
//     S4[c0 | c1 << 2 | c2 << 4 | c3 << 6] = (S'[sym(c0)] << 3) | (S'[sym(c1)] << 2) | (S'[sym(c2)] << 1) | S'[sym(c3)]
// of four consecutive symbols given by their two-bit codes c = (byte >> shift) & 3 — the operand of the step above,
// ready-made.  Per four text bytes: v_lshrrev + v_and (codes), v_dot4_u32_u8 (index), v_lshl_or (address), ONE gather,
// v_lshl_or (D), v_alignbit (hits) — 1.5 VALU ops and a quarter LDS gather per byte against 2.25 and one.  The few
// bytes of a run's first and last halves go a byte at a time through one shared copy of the 256 masks (at most four
// addresses per wave: no conflicts to speak of).
template <bool LONG, bool FOUR>  // LONG: m > 29, hits of the 29-byte prefix are verified
__global__ __launch_bounds__(kRunWaves * 64) void so_runs(ScanArgs a1, uint32_t run_len, uint64_t nruns, const BatchItem* __restrict__ batch)
{
    const ScanArgs a = pick_args(a1, batch);  // a pattern set in one grid: blockIdx.y = pattern (launch_batch)
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t m = a.m, w = m < kSoWindow ? m : kSoWindow;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    // FOUR: the text's two-bit codes, from the first words of the text's own allocation (TextCodes, kernels.hpp)
    const uint32_t four_shift = FOUR ? reinterpret_cast<const uint32_t*>(a.text - kFrontPad)[0] : 0u;
    const uint32_t four_symtab = FOUR ? reinterpret_cast<const uint32_t*>(a.text - kFrontPad)[1] : 0u;
    uint32_t* S = reinterpret_cast<uint32_t*>(smem);
    const RunIo io = swap_io(smem + 65536 + wave * kLineSlab, lane, run_len);
    const uint32_t sh = 29u - w;
    const uint32_t sentinel = (0xFFFFFFFFu << sh) & 0x1FFFFFFFu;  // mask of a byte outside the lane's range
    constexpr uint32_t kS1 = 65536 + kRunWaves * kLineSlab;  // FOUR: LDS offset of ONE copy of the masks
    {   // expand the 256 masks (FOUR: the 256 four-symbol values) to one copy per lane through a 1 KB staging area (wave 0's slab)
        uint32_t* stage = reinterpret_cast<uint32_t*>(smem + 65536);
        const uint32_t* Sg = reinterpret_cast<const uint32_t*>(a.blob + a.so_off);
        auto mask = [&](uint32_t c) { return (Sg[c] << sh) & 0x1FFFFFFFu; };
        if (threadIdx.x < 256) {
            if (FOUR) {
                auto of_code = [&](uint32_t code) { return mask((four_symtab >> (8u * (code & 3u))) & 0xFFu); };
                const uint32_t t = threadIdx.x;
                stage[t] = (of_code(t) << 3) | (of_code(t >> 2) << 2) | (of_code(t >> 4) << 1) | of_code(t >> 6);
                reinterpret_cast<uint32_t*>(smem + kS1)[t] = mask(t);
            } else {
                stage[threadIdx.x] = mask(threadIdx.x);
            }
        }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < 256 * 64; i += kRunWaves * 64) S[i] = stage[i >> 6];
    }
    // the perm result IS the LDS address: the table sits at LDS offset 0 (this kernel has no
    // static LDS, so the dynamic segment starts there); a poisoned count if that ever changes
    const uint32_t lane4 = lane * 4u;
    if ((uint32_t)(uintptr_t)(lds_u8_t*)smem != 0u) {
        if (threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(reinterpret_cast<unsigned long long*>(a.count), 1ull << 62);
        return;
    }
    __syncthreads();  // the only workgroup barrier: table visible

    uint32_t hits = 0;
    const uint64_t run_first = a.s_begin / run_len;
    const uint64_t nwaves = (uint64_t)gridDim.x * kRunWaves;
    const uint32_t nlines = (run_len + w - 1 + kRunLine - 1) / kRunLine;
    for (uint64_t g = first_group(nruns, 64, kRunWaves, wave); g * 64 < nruns; g += nwaves) {
        const uint8_t* const gbase = a.text + (run_first + g * 64) * run_len + io.loff;
        uint32_t blk[8];  // a block that lies entirely past the last run re-reads block 0 (loaded, never consumed)
#pragma unroll
        for (int i = 0; i < 8; ++i) blk[i] = g * 64 + 8 * i < nruns ? 8u * i * run_len : 0u;
        const uint64_t my = g * 64 + lane;
        const uint64_t seg = (run_first + my) * run_len;
        const uint64_t sa = seg > a.s_begin ? seg : a.s_begin;
        const uint64_t sb = seg + run_len < a.s_end ? seg + run_len : a.s_end;
        const bool owner = my < nruns && sa < sb;
        const uint32_t j0 = owner ? (uint32_t)(sa - seg) : 0u;
        const uint32_t jend = owner ? (uint32_t)(sb - seg) + w - 1 : 0u;

        uint4 n0, n1, n2, n3, n4, n5, n6, n7;
        LINE_FETCH(gbase, blk, 0u);
        uint32_t D = 0xFFFFFFFFu << sh;  // no prefix matched yet
        // LONG: the first prefix hit of a half waits here for wave_verify, as its offset in the run (0 = none: a
        // hit's offset is at least w); the run's text offset is recomputed there — nothing 64-bit stays live
        uint32_t parked_off = 0;
        // one 64-byte half of a line: the bytes [jb, jb + 64) of every run are in the slab
        auto half = [&](const uint32_t jb) {
            // hit mask of one 16-byte chunk (bit 15-q: a window ends at byte q)
            auto take_hits = [&](uint32_t base, uint32_t hm) {
                if (!LONG) hits += __popc(hm);
                else if (hm) {
                    const uint64_t r = so_long_hits(a.text + (run_first + my) * run_len, a.blob + w, m - w, base, hm, parked_off);
                    hits += (uint32_t)(r >> 32);
                    parked_off = (uint32_t)r;
                }
            };
            if (jb >= j0 && jb + 64u <= jend) {
                // the whole half is inside the run (all but a run's last): straight-line code, software-
                // pipelined by one 16-byte chunk: the 16 gathers of chunk c+1 are issued (a wave can have 15
                // LDS operations outstanding) before the masks of chunk c are combined and shifted into D
                uint4 v[4];
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4) v[c4] = run_piece(io, c4);
                uint32_t H[2] = {0xFFFFFFFFu, 0xFFFFFFFFu};
                if constexpr (FOUR) {
                    uint32_t t4[16];
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const uint4& vv = v[i >> 2];
                        const uint32_t d = (i & 3) == 0 ? vv.x : (i & 3) == 1 ? vv.y : (i & 3) == 2 ? vv.z : vv.w;
                        const uint32_t c = (d >> four_shift) & 0x03030303u;
                        const uint32_t idx = __builtin_amdgcn_udot4(c, 0x40100401u, 0u, false);  // c0 | c1 << 2 | c2 << 4 | c3 << 6
                        t4[i] = *(const lds_u32_t*)(size_t)((idx << 8) | lane4);
                    }
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        D = (D << 4) | t4[i];                                     // four steps of so.c:55
                        H[i >> 3] = __builtin_amdgcn_alignbit(H[i >> 3], D, 28);  // so.c:56 for the four bytes: bits 28..31
                    }
                } else {
                uint32_t s[2][16];
                auto gather16 = [&](const uint4& vv, uint32_t* out) {
                    const uint32_t d[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
                    for (int q = 0; q < 16; ++q)
                        out[q] = *(const lds_u32_t*)(size_t)__builtin_amdgcn_perm(d[q >> 2], lane4, 0x0c0c0400u + ((q & 3) << 8));
                };
                gather16(v[0], s[0]);
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4) {
                    if (c4 < 3) gather16(v[c4 + 1], s[(c4 + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                    const uint32_t* sc = s[c4 & 1];
                    uint32_t pr[8], t[4];
#pragma unroll
                    for (int k = 0; k < 8; ++k) pr[k] = lshl_or_now<1>(sc[2 * k], sc[2 * k + 1]);
#pragma unroll
                    for (int k = 0; k < 4; ++k) t[k] = lshl_or_now<2>(pr[2 * k], pr[2 * k + 1]);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        D = (D << 4) | t[k];                                        // four steps of so.c:55
                        H[c4 >> 1] = __builtin_amdgcn_alignbit(H[c4 >> 1], D, 28);  // so.c:56 for the four bytes: bits 28..31
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                }
                if (!LONG) {
                    hits += __popc(~H[0]) + __popc(~H[1]);
                } else if (__any((H[0] & H[1]) != 0xFFFFFFFFu)) {  // rare: ONE wave-uniform branch per half on the streaming path
#pragma unroll 1
                    for (int h = 0; h < 2; ++h) {
                        const uint32_t hm = ~(h ? H[1] : H[0]);
                        take_hits(jb + 32u * h, hm >> 16);
                        take_hits(jb + 32u * h + 16u, hm & 0xFFFFu);
                    }
                }
            } else {
#pragma unroll 1
                for (int c4 = 0; c4 < 4; ++c4) {
                    const uint32_t base = jb + 16u * c4;
                    if (base >= jend || base + 16 <= j0) continue;
                    const uint4 v = run_piece(io, c4);
                    const uint32_t d[4] = {v.x, v.y, v.z, v.w};
                    uint32_t H = 0xFFFFFFFFu;
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const uint32_t j = base + q;
                        // (an LDS address, not S1[.]: the select below must stay a select of VALUES)
                        uint32_t sv = FOUR ? *(const lds_u32_t*)(size_t)(kS1 + __builtin_amdgcn_perm(0u, d[q >> 2], 0x0c0c0c00u + (q & 3)) * 4u)
                                           : *(const lds_u32_t*)(size_t)__builtin_amdgcn_perm(d[q >> 2], lane4, 0x0c0c0400u + ((q & 3) << 8));
                        sv = (j >= j0 && j < jend) ? sv : sentinel;
                        D = (D << 1) | sv;                                // so.c:55
                        H = __builtin_amdgcn_alignbit(H, D << 3, 31);    // so.c:56: bit 28 = bit w-1 of the state
                    }
                    take_hits(base, ~H & 0xFFFFu);
                }
            }
            if (LONG && __any(parked_off != 0)) {  // wave-uniform point; at most one parked window per lane
                hits += wave_verify(parked_off != 0, a.text + (run_first + my) * run_len + parked_off, a.blob + w, m - w);
                parked_off = 0;
            }
        };
        for (uint32_t k = 0; k < nlines; ++k) {
            SWAP_LINE();
            RUN_PARK(io, n0, n2, n4, n6);
            half(k * kRunLine);
            RUN_PARK(io, n1, n3, n5, n7);
            if (k + 1 < nlines) LINE_FETCH(gbase, blk, (k + 1) * kRunLine);  // wave-uniform
            half(k * kRunLine + 64u);
        }
    }
    flush_hits(hits, a.count, smem);
}

// ---------------------------------------------------------------------------
// KMP over per-lane runs: the automaton's transition table (the failure function of kmp.c:27-41
// expanded on the host, one dependent LDS lookup per byte: st = delta[st][c]) with an ABSORBING accept
// row, behind the swap loader above.
//
// kmp_runs1 finds the halves in which an occurrence ended with a running maximum of the states (half
// a VALU op per byte on a kernel that is bound by instruction issue and by the latency of its lookup
// chain at about the same point).  Here the table itself remembers: every transition INTO the accept
// state w leads to an extra row Z whose entries all say Z.  A lane that comes out of a 64-byte half in
// Z saw an occurrence end there — one compare per half — and only then walks that half again from the
// state it had saved at the 16-byte chunk where it fell into Z, counting.  The counting walk uses the
// same table: Z = id(w) + 1 is the largest id, so min(next, id(w)) turns Z into the accept state's own
// row (whose entries are the real delta(w, .)) and next - min(..) is the hit — v_min_u32 + v_sad_u32.
// Per byte on the common path: v_perm_b32 (address: byte 1 = state, byte 0 = text byte), v_xor_b32
// (bank swizzle), ds_read_u8.
// State ids (api.cpp build_blob): fewer than 63 states: id(s) = 4s, the table ends with row Z = 4w+1;
// otherwise id(s) = rotl8(s, 2) with id(w) = 254, Z = 255 (the one state that would sit on 254, s = 191,
// takes the slot w left free).  w = m up to 254 bytes; longer patterns: the 62-byte prefix's automaton (kKmpPrefix), a
// prefix hit is parked and verified (wave_verify) — what so.c does with its 32-byte prefix.
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t sad_now(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t r;
    asm("v_sad_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// sixteen transitions, nothing else (Z absorbs)
__device__ __forceinline__ void kmp_chunk_fast(const uint4& v, uint32_t& st)
{
    const uint32_t d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int q = 0; q < 16; ++q) st = kmp_delta(d[q >> 2], st, q & 3);
}

// Sixteen transitions, four at a time where the whole WAVE is in LOW states.  Four bytes x on from state s the
// automaton of kmp.c:27-68 is in the state of the longest prefix of P that ends the text there: a prefix of at most
// four bytes is a suffix of x alone (4 if x is P[0..4), else 3 if its last three bytes are P[0..3), ... — four
// compares and four selects); a longer one, of L bytes, ends in x = P[L-4..L) and starts with a prefix of L-4 bytes
// that ended the text before x — s itself or one of its borders.  For a state s WITHOUT a border (kmpNext's chain from
// s leads straight to 0) that leaves L = s+4: one more compare, against the dword Q[s] = P[s..s+4) — a 256-byte table
// in LDS next to the transitions, indexed with the state's id 4s as the byte offset.  The host (api.cpp) finds the
// largest K such that no state 1..K has a border, K+4 < w (no occurrence can end inside the dword: counting stays
// with the lookups) and the ids up to K+4 are 4s; thr = 4K.  `low` — wave-uniform: every lane's state is at most K
// — holds for practically every dword on text over a large alphabet and on natural language (a lane beyond K has
// matched K+1 bytes of P), where z0 of the first version (all lanes in state 0) held for 60 % on rand128 and never on
// English; on small alphabets the form switches itself off (nfast, the dwords that went without lookups).
// every lane in a state 0..K = thr/4?  The ids of those states are 4s in both numberings, but an id at most thr need
// not be one of them: with 63 states or more id(s) = rotl8(s, 2) gives the states from 64 on the ids 1, 5, 9, ... —
// rotating the id right by two bits (32-bit) sends every id that is not a multiple of 4 beyond any thr.
__device__ __forceinline__ bool kmp_all_low(uint32_t st, uint32_t thr)
{
    return __ballot(__builtin_amdgcn_alignbit(st, st, 2) > (thr >> 2)) == 0;
}

struct KmpPrefix4 { uint32_t p4, p3, p2, p1; };  // P[0..4) as a dword, P[0..3) << 8, P[0..2) << 16, P[0] << 24

// the longest prefix of P, of at most four bytes, that ends the dword x (as a state id)
__device__ __forceinline__ uint32_t kmp_fresh4(uint32_t x, const KmpPrefix4& pf)
{
    uint32_t s = (x & 0xFF000000u) == pf.p1 ? 4u : 0u;       // the last byte is P[0]: state 1 (id 4)
    s = (x & 0xFFFF0000u) == pf.p2 ? 8u : s;                 // the last two are P[0..2): state 2
    s = (x & 0xFFFFFF00u) == pf.p3 ? 12u : s;                // the last three are P[0..3): state 3
    return x == pf.p4 ? 16u : s;                             // all four: state 4
}

// EXT = false: the form for the whole wave in state 0 (thr is 0: registers only); EXT = true: in the states 0..K
template <bool EXT>
__device__ __forceinline__ void kmp_chunk_skip4(const uint4& v, uint32_t& st, bool& low, const KmpPrefix4& pf, uint32_t qbase,
                                                uint32_t thr, uint32_t& nfast)
{
    const uint32_t d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (low) {
            const uint32_t x = d[k];
            if (EXT) {
                const uint32_t q = *(const lds_u32_t*)(size_t)(qbase + st);  // P[s..s+4), s = st / 4
                const uint32_t s = kmp_fresh4(x, pf);
                st = x == q ? st + 16u : s;                                  // the match went on: state s+4
            } else {
                st = kmp_fresh4(x, pf);
            }
            ++nfast;
            low = __ballot(st > thr) == 0;
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) st = kmp_delta(d[k], st, q);
            low = EXT ? kmp_all_low(st, thr) : __ballot(st != 0u) == 0;
        }
    }
}

// Sixteen transitions as FOUR table steps of four bytes each — a text of at most four distinct byte values
// (ScanArgs.four_shift; the kernel derives the table from the byte table when it starts, see there): the codes
// (c >> shift) & 3 of a dword's bytes make the index (v_lshrrev, v_and, v_dot4_u32_u8 — none of them on the chain), the
// step is one lookup in the row r = id + 2 of the state (the gaps of the byte table), whose entries are such rows again:
// v_lshl_or, v_xor and the LDS read per four bytes.  r: the state as that row, in and out.
__device__ __forceinline__ void kmp_chunk_four(const uint4& v, uint32_t& r, uint32_t shift)
{
    const uint32_t d[4] = {v.x, v.y, v.z, v.w};
    uint32_t idx[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)  // the first byte's code in bits 0-1, ... the fourth's in bits 6-7
        idx[k] = __builtin_amdgcn_udot4((d[k] >> shift) & 0x03030303u, 0x40100401u, 0u, false);
#pragma unroll
    for (int k = 0; k < 4; ++k) r = *(const lds_u8_t*)(size_t)(((r << 8) | idx[k]) ^ r);
}

// sixteen transitions, counting (MASK: collecting) the entries into Z; CHECK: only bytes j0 <= j < jend
template <bool CHECK, bool MASK>
__device__ __forceinline__ void kmp_chunk_count(const uint4& v, uint32_t j_base, uint32_t j0, uint32_t jend,
                                                uint32_t& st, uint32_t& hits, uint32_t idw)
{
    const uint32_t d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const uint32_t nx = kmp_delta(d[q >> 2], st, q & 3);
        const uint32_t real = nx < idw ? nx : idw;  // Z -> the accept state's own row
        if (CHECK) {
            const uint32_t j = j_base + q;
            const bool live = j >= j0 && j < jend;
            st = live ? real : st;
            if (MASK) hits |= (live && nx > idw) ? (1u << q) : 0u;
            else hits += live ? nx - real : 0u;
        } else {
            st = real;
            if (MASK) hits |= nx > idw ? (1u << q) : 0u;
            else hits = sad_now(nx, real, hits);
        }
    }
}

// Loader and fast forms, measured alternating on one box, ms per GiB of rand128 (round 2):
//   m <= 62:  swap loader, a lookup per byte 0.194-0.199; + the state-0 form 0.183-0.187; + the 0..K form where
//             state 0 does not cover the wave (rand32, English: 0.196 -> 0.188-0.190);
//   m = 64 .. 254 (full 64 KB table): half-line loader of kmp_runs1 0.206-0.210, swap 0.197-0.199, + forms 0.183-0.19;
//   m > 254 (PREFIX): half-line 0.207-0.213, swap 0.206-0.212, swap + the 0..K form 0.196-0.199 (the state-0 form
//             on its own made it slower: 0.218-0.232).
// FOUR — a separate INSTANTIATION, so that what it needs costs round 2's kernel nothing (as a run-time switch, together
// with a speculation for large alphabets that was dropped, it cost the English and rand32 cells 4-11 % against round 2's
// build: 124 VGPRs, a larger loop): the TEXT holds at most four distinct byte values (ScanArgs.four_shift, four_symtab)
// and the plan's window is at most 62 bytes (api.cpp build_blob: patterns over at most four symbols).  Row 4s + 2 of the
// table — the gaps of the byte table — then holds, for every index of four two-bit codes, the row the automaton is in
// four bytes on from state s: (id | 2), with the absorbing Z | 2 = 4w + 3 if an occurrence ended on the way.  The
// workgroup computes these rows itself before it starts, four lookups in the byte table per entry (0.5 us), so the
// codes are the text's own and the plan carries nothing for them.
template <bool PREFIX, bool FOUR>  // PREFIX: the automaton of the 62-byte prefix (m > 254; FOUR: m > 62); hits are verified
__global__ __launch_bounds__((FOUR ? kKmpFourWaves : kRunWaves) * 64) void kmp_runs(ScanArgs a1, uint32_t run_len, uint64_t nruns,
                                                           uint32_t dfa_off, const BatchItem* __restrict__ batch)
{
    const ScanArgs a = pick_args(a1, batch);  // a pattern set in one grid: blockIdx.y = pattern (launch_batch)
    constexpr int kW = FOUR ? kKmpFourWaves : kRunWaves;  // waves of the workgroup
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t m = a.m;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t w = PREFIX ? kKmpPrefix : m;  // length the automaton recognises
    const uint32_t idw = w < 63 ? 4 * w : 254u, Z = idw + 1;
    constexpr bool four = FOUR;
    // (+ two rows of that table: 4w + 2, the accept state's, and 4w + 3 = Z | 2, all Z: a lane that fell into Z stays there)
    const uint32_t table_bytes = (Z + 1 + (four ? 2u : 0u)) * 256;
    KmpPrefix4 pf;
    {
        const uint32_t p = *reinterpret_cast<const uint32_t*>(a.blob);  // P[0..4) (the pattern slot is zero-padded)
        pf.p4 = p;
        pf.p3 = p << 8;
        pf.p2 = p << 16;
        pf.p1 = p << 24;
    }
    const uint32_t qbase = table_bytes;  // Q[s] = P[s..s+4) for the states kmp_chunk_skip4 covers (256 bytes)
    const uint32_t stored = (w < 63 ? w + 1 : 256u) * 256u;  // the blob's rows (tables.cpp kmp_runs_tables)
    const uint32_t thr = *reinterpret_cast<const uint32_t*>(a.blob + dfa_off + stored + 256);
    // FOUR: the text's two-bit codes, from the first words of the text's own allocation (TextCodes, kernels.hpp)
    const uint32_t shift4 = FOUR ? reinterpret_cast<const uint32_t*>(a.text - kFrontPad)[0] : 0u;
    const uint32_t four_symtab = FOUR ? reinterpret_cast<const uint32_t*>(a.text - kFrontPad)[1] : 0u;
    uint8_t* const slabs = smem + table_bytes + kKmpQBytes;
    const RunIo io = swap_io(slabs + wave * kLineSlab, lane, run_len);
    {
        const uint4* g = reinterpret_cast<const uint4*>(a.blob + dfa_off);
        uint4* t = reinterpret_cast<uint4*>(smem);
        if (w < 63) {
            // the blob holds the rows of the states 0..w one after the other: row s goes to row 4s (the rows between
            // are never addressed), row Z is filled here, Q follows the table
            for (uint32_t i = threadIdx.x; i < (w + 1) * 16; i += kW * 64) t[(i >> 4) * 64 + (i & 15u)] = g[i];
            const uint32_t z4 = Z * 0x01010101u;
            if (threadIdx.x < 16) t[Z * 16 + threadIdx.x] = make_uint4(z4, z4, z4, z4);
            else if (threadIdx.x < 32) t[table_bytes / 16 + threadIdx.x - 16] = g[stored / 16 + threadIdx.x - 16];
            if (four && threadIdx.x >= 32 && threadIdx.x < 48) {  // row Z | 2: a lane that fell into Z stays there
                const uint32_t zz = (Z | 2u) * 0x01010101u;
                t[(Z + 2) * 16 + threadIdx.x - 32] = make_uint4(zz, zz, zz, zz);
            }
        } else {
            for (uint32_t i = threadIdx.x; i < (table_bytes + 256) / 16; i += kW * 64) t[i] = g[i];
        }
    }
    // the perm result IS the LDS address: the table sits at LDS offset 0 (no static LDS in
    // this kernel, so the dynamic segment starts there); a poisoned count if that ever changes
    if ((uint32_t)(uintptr_t)(lds_u8_t*)smem != 0u) {
        if (threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(reinterpret_cast<unsigned long long*>(a.count), 1ull << 62);
        return;
    }
    __syncthreads();  // table visible
    if (four) {  // the rows 4s + 2: four steps of the byte table for every index of four codes
        for (uint32_t i = threadIdx.x; i < (w + 1) * 256u; i += kW * 64) {
            const uint32_t s = i >> 8, idx = i & 255u, r = 4u * s + 2u;
            uint32_t st = 4u * s;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t c = (four_symtab >> (8u * ((idx >> (2 * j)) & 3u))) & 0xFFu;
                st = *(const lds_u8_t*)(size_t)(((st << 8) | c) ^ st);  // kmp_delta
            }
            smem[(r << 8) | ((idx ^ r) & 255u)] = (uint8_t)(st | 2u);  // Z | 2 if an occurrence ended on the way
        }
        __syncthreads();
    }

    uint32_t hits = 0;
    const uint64_t run_first = a.s_begin / run_len;  // runs are cut on absolute offsets
    const uint64_t nwaves = (uint64_t)gridDim.x * kW;
    const uint32_t nlines = (run_len + w - 1 + kRunLine - 1) / kRunLine;
    for (uint64_t g = first_group(nruns, 64, kW, wave); g * 64 < nruns; g += nwaves) {
        const uint8_t* const gbase = a.text + (run_first + g * 64) * run_len + io.loff;
        uint32_t blk[8];  // a block that lies entirely past the last run re-reads block 0 (loaded, never consumed)
#pragma unroll
        for (int i = 0; i < 8; ++i) blk[i] = g * 64 + 8 * i < nruns ? 8u * i * run_len : 0u;
        const uint64_t my = g * 64 + lane;
        const uint64_t seg = (run_first + my) * run_len;
        const uint64_t sa = seg > a.s_begin ? seg : a.s_begin;
        const uint64_t sb = seg + run_len < a.s_end ? seg + run_len : a.s_end;
        const bool owner = my < nruns && sa < sb;
        const uint32_t j0 = owner ? (uint32_t)(sa - seg) : 0u;
        const uint32_t jend = owner ? (uint32_t)(sb - seg) + w - 1 : 0u;  // bytes [j0, jend) of the run can end an occurrence

        uint4 n0, n1, n2, n3, n4, n5, n6, n7;
        LINE_FETCH(gbase, blk, 0u);
        uint32_t st = 0;
        // wave-uniform: which form walks the whole halves — 1: four bytes at a time while every lane is in state 0
        // (registers only; most dwords on a large alphabet), 2: while every lane is in a state 0..K (one Q lookup per
        // dword; natural language, medium alphabets — not worth trying with K < 4), 0: a lookup per byte.  A half in
        // which a form covered fewer than 6 of the 16 dwords hands over to the next one; tried again from the top every
        // 8 lines.  The prefix automaton starts with form 2 (measured, above).
        const uint32_t mode0 = four ? 5u : w < 5 ? 0u : !PREFIX ? 1u : thr >= 16u ? 2u : 0u;
        uint32_t mode = mode0;
        bool dense = false;   // wave-uniform: many lanes saw an occurrence end in the last whole half
        bool parked = false;  // PREFIX: first unverified prefix hit of this step
        const uint8_t* parked_at = a.text;
        auto half = [&](const uint32_t jb) {
            // one 16-byte chunk, counting; returns whether an occurrence ended in it
            // A chunk outside [j0, jend) — the text's last run ends early, the lanes of the last group may have no run
            // at all — must cost nothing, not even its slab read: such a lane makes its wave run this path next to the
            // straight one in every half, and with the read in front of the test that one wave ended the kernel 7-10 %
            // late (measured: 2^30 bytes against 2^30 - 26 runs; m = 4096, whose last run has ONE start, against 1024).
            auto careful = [&](uint32_t q, bool whole) -> bool {
                const uint32_t j = jb + 16u * q;
                if (!whole && !(j < jend && j + 16 > j0)) return false;
                const uint4 v = *reinterpret_cast<const uint4*>(io.rd + ((16u * q) ^ io.rswz));
                if (!PREFIX) {
                    const uint32_t h0 = hits;
                    if (whole) kmp_chunk_count<false, false>(v, j, j0, jend, st, hits, idw);
                    else kmp_chunk_count<true, false>(v, j, j0, jend, st, hits, idw);
                    return hits != h0;
                } else {
                    uint32_t hm = 0;
                    if (whole) kmp_chunk_count<false, true>(v, j, j0, jend, st, hm, idw);
                    else kmp_chunk_count<true, true>(v, j, j0, jend, st, hm, idw);
                    const bool seen = hm != 0;
                    while (hm) {  // the prefix ends at byte j+b: verify P[w..m)
                        const uint32_t b = __builtin_ctz(hm);
                        hm &= hm - 1;
                        const uint8_t* rest = a.text + seg + j + b + 1;  // = text + start + w
                        if (!parked) {
                            parked = true;
                            parked_at = rest;
                        } else {
                            hits += global_equal(rest, a.blob + w, m - w);
                        }
                    }
                    return seen;
                }
            };
            if (jb >= j0 && jb + 64u <= jend) {  // the whole half is inside the run
                bool seen = false;
                if (!dense) {
                    uint32_t at[4];  // state before each 16-byte chunk
                    if (mode == 1) {
                        bool low = __ballot(st != 0u) == 0;
                        uint32_t nfast = 0;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            at[q] = st;
                            kmp_chunk_skip4<false>(run_piece(io, q), st, low, pf, qbase, 0u, nfast);
                        }
                        if (nfast < 6) mode = thr >= 16u ? 2u : 0u;
                    } else if (FOUR && mode == 5) {
                        uint32_t r = st | 2u;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            at[q] = r & ~2u;
                            kmp_chunk_four(run_piece(io, q), r, shift4);
                        }
                        st = r & ~2u;
                    } else if (mode == 2) {
                        bool low = kmp_all_low(st, thr);
                        uint32_t nfast = 0;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            at[q] = st;
                            kmp_chunk_skip4<true>(run_piece(io, q), st, low, pf, qbase, thr, nfast);
                        }
                        if (nfast < 6) mode = 0u;
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            at[q] = st;
                            kmp_chunk_fast(run_piece(io, q), st);
                        }
                    }
                    seen = st == Z;
                    if (__any(seen)) {
                        if (seen) {
                            // the chunk in which the lane fell into Z: walk on from there, counting
                            const uint32_t q0 = at[1] == Z ? 0u : at[2] == Z ? 1u : at[3] == Z ? 2u : 3u;
                            st = q0 == 0 ? at[0] : q0 == 1 ? at[1] : q0 == 2 ? at[2] : at[3];
#pragma unroll 1
                            for (uint32_t q = q0; q < 4; ++q) careful(q, true);
                        }
                    }
                } else {
#pragma unroll 1
                    for (uint32_t q = 0; q < 4; ++q) seen |= careful(q, true);
                }
                // where occurrences are frequent (short patterns, small alphabets) walking twice
                // costs more than it saves: the wave counts directly while an eighth of its lanes
                // saw one in the last half
                dense = __popcll(__ballot(seen)) >= 8;
            } else if (jb < jend && jb + 64u > j0) {  // an end of the run lies in this half
#pragma unroll 1
                for (uint32_t q = 0; q < 4; ++q) {
                    const uint32_t j = jb + 16u * q;
                    careful(q, j >= j0 && j + 16 <= jend);
                }
            }
            if (PREFIX && __any(parked)) {  // wave-uniform point: at most one parked hit per lane
                hits += wave_verify(parked, parked_at, a.blob + w, m - w);
                parked = false;
            }
        };
        for (uint32_t k = 0; k < nlines; ++k) {
            if ((k & 7u) == 7u) mode = mode0;
            SWAP_LINE();
            RUN_PARK(io, n0, n2, n4, n6);
            half(k * kRunLine);
            RUN_PARK(io, n1, n3, n5, n7);
            if (k + 1 < nlines) LINE_FETCH(gbase, blk, (k + 1) * kRunLine);  // wave-uniform
            half(k * kRunLine + 64u);
        }
    }
    flush_hits(hits, a.count, smem);
}

#ifdef SMARTGPU_AB
// The superseded kernels kept for A/B measurements (so_scan, so_runs64, so_runs1, kmp_scan, kmp_links_runs,
// kmp_runs1 and their loaders): only in the A/B build (make AB=1 -> libsmartgpu_ab.so), selected by smartgpu_tune.
#include "kernels_ab.inc"
#endif

// ---------------------------------------------------------------------------
// EPSM — packed matching  (reference: src/algos/epsm.c; its SSE regimes —
// broadcast compare, mpsadbw 4-byte filter, hashed 8-byte blocks — map to one
// VALU scheme here: compare the first F = min(m,16) pattern bytes, packed as up
// to four masked dwords, at EVERY alignment; verify the rest only on a hit).
//
// Each lane takes 16 consecutive start positions per row straight from
// registers: 32 text bytes (its own 16 and the next 16) give the text dword at
// each of its 16 byte offsets via v_alignbyte_b32.  Deeper fingerprint dwords
// are compared only when some lane of the wave still has a candidate (ballot);
// m > 16 verifies bytes 16.. from memory.  Hits are popcounts of the per-lane
// candidate masks.  No LDS, no table beyond the 4-dword fingerprint.
// ROWS rows (ROWS * 4 KiB per workgroup) are loaded before any is processed so
// that enough bytes are in flight per CU to cover HBM latency.
// ---------------------------------------------------------------------------
struct EpsmFp { uint32_t f0, f1, f2, f3, k0, k1, k2, k3, nd, m; };

// text dword at byte offset x (compile-time after unrolling) of the 8-dword window d[]
#define SG_W(x) (((x) & 3) == 0 ? d[(x) >> 2] \
                                : __builtin_amdgcn_alignbyte(d[((x) >> 2) + 1], d[(x) >> 2], (x) & 3))

// MODE 0: some fingerprint dword is partial (m < 16, m % 4 != 0) -> masked compares;
// MODE 1: whole dwords only; MODE 2: m > 16, four whole dwords + bytes 16.. verified in memory
// MASK: return the surviving offsets in `pending` instead of counting them (packed_find)
template <int MODE, bool MASK = false>
static __device__ __forceinline__ uint32_t epsm_row(const ScanArgs& a, const EpsmFp& fp,
                                                    const uint4& A, const uint4& Bv, uint64_t p0,
                                                    uint32_t& pending, bool overlap_lane)
{
    const uint32_t d[8] = {A.x, A.y, A.z, A.w, Bv.x, Bv.y, Bv.z, Bv.w};
    // offsets k with p0+k inside [s_begin, s_end)
    uint32_t cand = overlap_lane ? 0u : 0xFFFFu;
    if (p0 < a.s_begin || p0 + 16 > a.s_end) {
        const uint64_t lo64 = a.s_begin > p0 ? a.s_begin - p0 : 0;
        const uint64_t hi64 = a.s_end > p0 ? a.s_end - p0 : 0;
        const uint32_t lo = lo64 > 16 ? 16u : (uint32_t)lo64;
        const uint32_t hi = hi64 > 16 ? 16u : (uint32_t)hi64;
        cand = (hi > lo && !overlap_lane) ? (((1u << hi) - 1u) & ~((1u << lo) - 1u)) : 0u;
    }
    constexpr bool VERIFY = MODE == 2;
#define SG_EQ(x, kk, ff) (MODE != 0 ? (SG_W(x) == (ff)) : ((SG_W(x) & (kk)) == (ff)))
    const uint32_t nd = VERIFY ? 4u : fp.nd;
    {
        uint32_t eq = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) eq |= SG_EQ(k, fp.k0, fp.f0) ? (1u << k) : 0u;
        cand &= eq;
    }
    if (nd > 1 && __any(cand != 0)) {
        uint32_t eq = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) eq |= SG_EQ(k + 4, fp.k1, fp.f1) ? (1u << k) : 0u;
        cand &= eq;
        if (nd > 2 && __any(cand != 0)) {
            eq = 0;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                bool e2 = SG_EQ(k + 8, fp.k2, fp.f2);
                if (nd > 3) e2 = e2 && SG_EQ(k + 12, fp.k3, fp.f3);
                eq |= e2 ? (1u << k) : 0u;
            }
            cand &= eq;
        }
    }
#undef SG_EQ
    if (MODE == 2 || MASK) {  // the caller verifies bytes 16.. of the survivors (epsm_verify) / emits them
        pending = cand;
        return 0;
    }
    return __popc(cand);
}
#undef SG_W

// m > 16: candidates that matched the 16-byte fingerprint.  The lowest candidate of
// every lane goes to wave_verify, further ones (rare) are checked by the lane itself.
static __device__ __attribute__((noinline)) uint32_t epsm_verify(const uint8_t* text, const uint8_t* blob,
                                                                 uint32_t m, uint32_t cand, uint64_t p0)
{
    // out of line on purpose: inlined, its control flow pushes the streaming loop of
    // packed_scan over the SGPR budget (spills into the hot path, -12 % measured)
    const uint32_t len = m - 16;
    uint32_t c = cand;
    const bool has = c != 0;
    const uint32_t k0 = has ? __builtin_ctz(c) : 0u;
    c &= c - 1;
    cand &= ~(1u << k0);
    while (c) {
        const uint32_t k = __builtin_ctz(c);
        c &= c - 1;
        if (!global_equal(text + p0 + k + 16, blob + 16, len)) cand &= ~(1u << k);
    }
    return __popc(cand) + wave_verify(has, text + p0 + k0 + 16, blob + 16, len);
}

// ALGO only tags the instantiation (rocprofv3 shows packed_scan<256, 4, 5, ..> for
// EPSM and packed_scan<256, 4, 0, ..> for Horspool's short-pattern regime).
template <int THREADS, int ROWS, int ALGO, int MODE, int POLICY>
__global__ __launch_bounds__(THREADS) void packed_scan(ScanArgs a1, uint64_t row_first,
                                                       uint64_t nrows, const BatchItem* __restrict__ batch)
{
    const ScanArgs a = pick_args(a1, batch);  // a pattern set in one grid: blockIdx.y = pattern (launch_batch)
    // POLICY 0: A non-temporal, B cached (default); 1: both cached; 3: one nt load + shuffle.
    // (Both loads nt measured 62-67 %: the second load must find the line still cached.  A
    // ballot/SGPR formulation of the first-dword test measured 59-73 %: scalar-unit bound.)
    // Also measured and dropped (profiles/r01 session p): completing the few survivors of a dword in
    // memory instead of testing the next dword at all alignments (English 55-60 % -> 40-57 %: the
    // cached loads stall every row), a sparse pre-pass (running minimum of text
    // dword ^ f0, 2.25 instead of 3.25 VALU ops per alignment), two steps of loads in flight in
    // registers, and capping the resident workgroups through an LDS allocation — all within noise of
    // 76-80 %; the VALU is 63 % busy, the waves wait on memory half of their time (PMC).
    // Session t, all measured and dropped:
    //  * the same rows over hor_scan's data path (16 KB tiles staged in LDS with non-temporal loads, every
    //    byte fetched once, four workgroups per CU): rand128 68-82 % (here 70-81 %), English 43-60 % (47-74 %),
    //    rand4 52-57 % (63-66 %) — with candidates in most rows the few resident waves cannot hide the deeper
    //    fingerprint dwords and the verification;
    //  * natural language: a pattern whose first dword is frequent ("And ", "of t") runs two or three dword
    //    stages in most rows (42-55 % against 65-74 % for other English patterns).  Comparing the rarest
    //    dword first — picked from the pattern's own symbol counts, or from byte counts of the text — moved
    //    single patterns both ways (byte counts know nothing of "\nAnd"); comparing the XOR of the four
    //    dwords first (the funnel shift is linear over XOR: +1 op per alignment) lifted the worst patterns
    //    to 56-72 % but cost the median English pattern 2-4 points and rand32 8; choosing the dword at run
    //    time inside this loop cost EVERY pattern 20 % (rand128 80 % -> 62 %).
    constexpr bool NTA = POLICY != 1, NTB = false;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];  // 128 bytes: flush_hits
    const uint32_t* fpw = reinterpret_cast<const uint32_t*>(a.blob + a.fp_off);
    EpsmFp fp;
    fp.f0 = fpw[0]; fp.f1 = fpw[1]; fp.f2 = fpw[2]; fp.f3 = fpw[3];
    fp.k0 = fpw[4]; fp.k1 = fpw[5]; fp.k2 = fpw[6]; fp.k3 = fpw[7];
    fp.m = a.m;
    fp.nd = (a.m >= 13) ? 4 : (a.m + 3) / 4;  // fingerprint dwords

    uint32_t hits = 0;
    // POLICY 3: a wave-row is 63*16 = 1008 start positions; lane i loads the 16 bytes at
    // row + 16*i ONCE (non-temporal) and takes the next 16 bytes from lane i+1 by a
    // cross-lane shuffle; lane 63 only supplies the overlap into the next wave-row.
    // Other policies: a row is THREADS*16 offsets and every lane loads 32 bytes.
    constexpr bool SHUF = POLICY == 3;
    constexpr uint32_t ROW_BYTES = SHUF ? (THREADS / 64) * 1008u : THREADS * 16u;
    const uint32_t in_row = SHUF ? (threadIdx.x >> 6) * 1008u + (threadIdx.x & 63u) * 16u : threadIdx.x * 16u;
    // a workgroup takes ROWS consecutive rows per step
    for (uint64_t g = (uint64_t)blockIdx.x * ROWS; g < nrows; g += (uint64_t)gridDim.x * ROWS) {
        uint4 A[ROWS], B[ROWS];
#pragma unroll
        for (int j = 0; j < ROWS; ++j) {
            const uint64_t r = g + j < nrows ? g + j : nrows - 1;  // clamp, ignored below
            const uint8_t* src = a.text + (row_first + r) * ROW_BYTES + in_row;
            if (SHUF) {
                A[j] = ld_stream16(src);
            } else {
                // A is this lane's own 16 bytes; B re-reads the next lane's 16 bytes
                A[j] = NTA ? ld_stream16(src) : *reinterpret_cast<const uint4*>(src);
                B[j] = NTB ? ld_stream16(src + 16) : *reinterpret_cast<const uint4*>(src + 16);
            }
        }
        if (SHUF) {
#pragma unroll
            for (int j = 0; j < ROWS; ++j) {
                B[j].x = __shfl_down(A[j].x, 1, 64);
                B[j].y = __shfl_down(A[j].y, 1, 64);
                B[j].z = __shfl_down(A[j].z, 1, 64);
                B[j].w = __shfl_down(A[j].w, 1, 64);
            }
        }
        uint32_t pend[ROWS];
#pragma unroll
        for (int j = 0; j < ROWS; ++j) {
            pend[j] = 0;
            if (g + j < nrows)
                hits += epsm_row<MODE>(a, fp, A[j], B[j], (row_first + g + j) * ROW_BYTES + in_row, pend[j],
                                       SHUF && (threadIdx.x & 63u) == 63u);
        }
        if (MODE == 2) {
            uint32_t any_pend = 0;
#pragma unroll
            for (int j = 0; j < ROWS; ++j) any_pend |= pend[j];
            if (__any(any_pend != 0)) {  // rare: one copy of the verification code, rows by select
#pragma unroll 1
                for (int j = 0; j < ROWS; ++j) {
                    uint32_t c = pend[0];
#pragma unroll
                    for (int q = 1; q < ROWS; ++q)
                        if (j == q) c = pend[q];
                    if (__any(c != 0))
                        hits += epsm_verify(a.text, a.blob, a.m, c, (row_first + g + j) * ROW_BYTES + in_row);
                }
            }
        }
    }
    flush_hits(hits, a.count, smem);
}

// Occurrence POSITIONS (an extension: the reference only counts, define.h:33).  The packed
// matcher with an output stage: the offsets that survive the fingerprint (and, for m > 16, the
// comparison of bytes 16.. in memory) are appended to `out`.  One atomic per wave and row that
// has hits: lanes' counts are prefix-summed inside the wave, lane 0 reserves the wave's span of
// the output.  `out_count` always receives the total; entries past `cap` are dropped.
template <int THREADS, int MODE>
__global__ __launch_bounds__(THREADS) void packed_find(ScanArgs a, uint64_t row_first, uint64_t nrows,
                                                       unsigned long long* out, unsigned long long cap)
{
    const uint32_t* fpw = reinterpret_cast<const uint32_t*>(a.blob + a.fp_off);
    EpsmFp fp;
    fp.f0 = fpw[0]; fp.f1 = fpw[1]; fp.f2 = fpw[2]; fp.f3 = fpw[3];
    fp.k0 = fpw[4]; fp.k1 = fpw[5]; fp.k2 = fpw[6]; fp.k3 = fpw[7];
    fp.m = a.m;
    fp.nd = (a.m >= 13) ? 4 : (a.m + 3) / 4;
    constexpr uint32_t ROW_BYTES = THREADS * 16u;
    const uint32_t lane = threadIdx.x & 63u;
    for (uint64_t g = blockIdx.x; g < nrows; g += gridDim.x) {
        const uint64_t p0 = (row_first + g) * ROW_BYTES + threadIdx.x * 16u;
        const uint8_t* src = a.text + p0;
        const uint4 A = ld_stream16(src);
        const uint4 B = *reinterpret_cast<const uint4*>(src + 16);
        uint32_t cand = 0;
        epsm_row<MODE, true>(a, fp, A, B, p0, cand, false);
        if (MODE == 2) {  // m > 16: bytes 16.. of every survivor
            uint32_t c = cand;
            while (c) {
                const uint32_t k = __builtin_ctz(c);
                c &= c - 1;
                if (!global_equal(a.text + p0 + k + 16, a.blob + 16, a.m - 16)) cand &= ~(1u << k);
            }
        }
        if (!__any(cand != 0)) continue;
        // wave-wide exclusive prefix sum of the lanes' hit counts
        const uint32_t mine = __popc(cand);
        uint32_t incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = __shfl_up(incl, d, 64);
            if (lane >= (uint32_t)d) incl += up;
        }
        const uint32_t total = __shfl(incl, 63, 64);
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(a.count, (unsigned long long)total);
        base = ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(base >> 32)) << 32) |
               __builtin_amdgcn_readfirstlane((uint32_t)base);
        unsigned long long slot = base + (incl - mine);
        while (cand) {
            const uint32_t k = __builtin_ctz(cand);
            cand &= cand - 1;
            if (slot < cap) out[slot] = p0 + k;
            ++slot;
        }
    }
}

// ---------------------------------------------------------------------------
// corpus kernels
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// dst[i] = corpus byte (off+i); one thread produces one aligned 8-byte corpus word
// (one 8-byte store when the word lies wholly inside the request and off%8==0).
__global__ __launch_bounds__(256) void generate_text(uint8_t* dst, uint64_t seed, uint32_t sigma,
                                                     uint64_t off, uint64_t n)
{
    const uint64_t w_first = off >> 3, w_last = (off + n + 7) >> 3;  // corpus words touched
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const bool pow2 = (sigma & (sigma - 1)) == 0;
    const bool aligned = (off & 7) == 0;
    for (uint64_t wi = w_first + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; wi < w_last;
         wi += stride) {
        const uint64_t x = splitmix64(seed + wi);
        uint64_t y;
        if (pow2) {
            y = x & (0x0101010101010101ull * (uint64_t)(sigma - 1));
        } else {
            y = 0;
#pragma unroll
            for (int b = 0; b < 8; ++b)
                y |= (uint64_t)(((uint32_t)(x >> (8 * b)) & 0xFFu) % sigma) << (8 * b);
        }
        const uint64_t j0 = wi << 3;  // corpus offset of byte 0 of this word
        if (aligned && j0 + 8 <= off + n) {
            *reinterpret_cast<uint64_t*>(dst + (j0 - off)) = y;
        } else {
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const uint64_t j = j0 + b;
                if (j >= off && j < off + n) dst[j - off] = (uint8_t)(y >> (8 * b));
            }
        }
    }
}

// dst[i] = unit[(phase + i) % unit_len]
__global__ __launch_bounds__(256) void tile_fill(uint8_t* dst, const uint8_t* unit,
                                                 uint64_t unit_len, uint64_t phase, uint64_t n)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        dst[i] = unit[(phase + i) % unit_len];
}

// Which byte values occur in the text (taken once, when a text is created — api.cpp text_alphabet): every workgroup
// marks them in LDS and ORs its 256 bits into out[8].
__global__ __launch_bounds__(256) void text_alphabet(const uint8_t* text, uint64_t n, uint32_t* out)
{
    __shared__ uint32_t seen[256];
    seen[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t n16 = n / 16, stride = (uint64_t)gridDim.x * 256;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) {
        const uint4 v = ld_stream16(text + 16 * i);
        const uint32_t d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 16; ++q) seen[(d[q >> 2] >> (8 * (q & 3))) & 0xFFu] = 1;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 15u)) seen[text[16 * n16 + threadIdx.x]] = 1;
    __syncthreads();
    const uint64_t bits = __ballot(seen[threadIdx.x] != 0);  // wave w of the workgroup: byte values 64w .. 64w+63
    if ((threadIdx.x & 63u) == 0) {
        if ((uint32_t)bits) atomicOr(out + 2 * (threadIdx.x >> 6), (uint32_t)bits);
        if ((uint32_t)(bits >> 32)) atomicOr(out + 2 * (threadIdx.x >> 6) + 1, (uint32_t)(bits >> 32));
    }
}

hipError_t launch_text_alphabet(const uint8_t* text, uint64_t n, uint32_t* out, int num_cus, hipStream_t stream)
{
    const uint64_t want = (n / 16 + 255) / 256;
    const uint32_t grid = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(want, (uint64_t)num_cus * 8));
    hipLaunchKernelGGL(text_alphabet, dim3(grid), dim3(256), 0, stream, text, n, out);
    return hipGetLastError();
}

// Streaming-read probe: the practical HBM read ceiling of this device for the
// access pattern the scan kernels use (coalesced 16 B/lane, 8 loads in flight per
// lane, every byte read once).  XOR-folds the text so the loads cannot be elided.
__global__ __launch_bounds__(256) void probe_read(const uint8_t* text, uint64_t n16,
                                                  unsigned long long* sink)
{
    const uint4* p = reinterpret_cast<const uint4*>(text);
    uint4 acc = {0, 0, 0, 0};
    const uint64_t stride = (uint64_t)gridDim.x * 256 * 8;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 * 8 + threadIdx.x; i < n16; i += stride) {
        uint4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k)
            v[k] = i + k * 256 < n16 ? ld_stream16(reinterpret_cast<const uint8_t*>(p + i + k * 256)) : uint4{0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 8; ++k) { acc.x ^= v[k].x; acc.y ^= v[k].y; acc.z ^= v[k].z; acc.w ^= v[k].w; }
    }
    const uint32_t f = acc.x ^ acc.y ^ acc.z ^ acc.w;
    if (f == 0x9E3779B9u) atomicAdd(sink, 1ull);  // practically never; keeps the loads live
}

hipError_t launch_probe_read(const uint8_t* text, uint64_t n, unsigned long long* sink, int num_cus,
                             hipStream_t stream)
{
    hipLaunchKernelGGL(probe_read, dim3((uint32_t)num_cus * 8), dim3(256), 0, stream, text, n / 16, sink);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
namespace {

struct TileRange { uint64_t first; uint32_t count; };
// A pattern set in ONE grid: while set, every scan launch uses gridDim.y = count and hands the kernels the
// device array of per-pattern arguments (they take argument set blockIdx.y instead of the by-value one).
struct BatchCtx { const BatchItem* items; uint32_t count; };
thread_local BatchCtx g_batch = {nullptr, 1};

// tiles of `tb` absolute offsets intersecting [lo, hi)
TileRange tiles_for(uint64_t lo, uint64_t hi, uint64_t tb)
{
    if (hi <= lo) return {0, 0};
    const uint64_t first = lo / tb, last = (hi - 1) / tb;
    return {first, (uint32_t)(last - first + 1)};
}

uint32_t r16(uint32_t x) { return (x + 15u) & ~15u; }

// Workgroups per CU of the LDS-tile skip kernels.  FOUR (64 KB of tiles in flight per CU), not
// the eight or nine the LDS would hold, when the pattern promises a pure streaming scan
// (a.sparse, api.cpp): measured on 1, 1.37 and 4 GiB of rand128, HOR m=32 runs at 85-87 % of
// 8 TB/s with 4, 78-82 % with 8, 75 % with 6, 82 % with 16 (two rounds) — profiles/r01/
// o_wgs_per_cu.log; BM and BNDM follow the same curve.  Where lanes spend their time verifying
// (English text: HOR m=64 47 % with 8, 40 % with 4) the extra waves pay: EIGHT.
// Short windows (8 <= m < 16) of such patterns: FIVE (HOR m = 8..13: 75-83 % with 5, 74-83 % with 6, 68-83 %
// with 4, 71-74 % with 8).  bm_scan, with its larger tables: THREE for m >= 16 (83-84 % against 79-82 % with 4).
static int tile_wgs(const ScanArgs& a, bool bm = false)
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

}  // namespace

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
        case 3: return value == 0 || value == 5;                 // superseded KMP kernels (5: kmp_runs without its four-byte table)
        case 6: return value == 0 || value == 5;                 // superseded SO kernels (5: so_runs without the four-symbol table)
        case 7: return value == 0;                               // packed load policies
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

// tile shapes (threads, bytes per lane)
constexpr int kHorT = 256, kHorL = 64;
constexpr int kBmT = 256, kBmL = 64;
constexpr int kBmBusyT = 128;  // bm_scan where windows survive (English, small alphabets): two-wave workgroups, 12 per CU (launch_scan)
constexpr int kBndmT = 256, kBndmL = 64;
constexpr int kBndmBusyT = 128;  // bndm_scan where windows survive: two-wave workgroups (launch_scan)
#ifdef SMARTGPU_AB
constexpr int kSoT = 256, kSoL = 80;  // so_scan
#endif
constexpr int kEpsmT = 256;

static int hor_regime(uint32_t m, int algo = SMARTGPU_HOR);

const char* scan_kernel_name(int algo, uint32_t m, bool prefer_packed, bool so_masks)
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
        case SMARTGPU_BNDM: return (pk || (m <= packed_max_m(SMARTGPU_BNDM) && g_tune[0] != 1)) ? "packed_scan" : algo == SMARTGPU_SBNDM ? "sbndm_scan" : "bndm_scan";
        case SMARTGPU_EPSM: return "packed_scan";
    }
    return "?";
}

// KMP runs.  Run length: re-scan overhead (m-1)/L <= 1/8 when the text is long enough,
// never above 1/2, short enough to give every CU ~16 waves of runs, multiple of 64.
// Run length for the runs kernels: every wave should get the same number of groups (`per_group`
// runs each), or the slowest wave sets the kernel time (12 waves/CU on 4096 groups: 67 %).
// Picks the smallest k such that span / (k * nwaves groups) gives runs of at most `lmax` bytes,
// then grows L in 64-byte steps until the runs cut on absolute offsets fit k * nwaves groups.
// Runs shorter than `lmin` (small texts) are not worth balancing: L = lmin.
static uint64_t balanced_run_len(uint64_t s_begin, uint64_t s_end, uint64_t per_group, uint64_t nwaves,
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
// contiguous share of the groups of 64 runs).  One pattern: as many workgroups as there are groups, up to one per CU —
// a small text is spread over the chip, one wave per CU.  A pattern set in one grid (gridDim.y patterns, SMART's -pset
// loop on its 1 MiB texts): every workgroup copies its pattern's table (up to 64 KB) before it starts, so with enough
// patterns to fill the chip the groups are packed 16 to a workgroup — 500 patterns x 128 groups of 128-byte runs:
// 4000 workgroups with all waves busy instead of 64000 with one (KMP 4.3 -> 0.6 us per pattern, measured).
static uint64_t runs_grid(uint64_t nruns, int num_cus, int waves = kRunWaves)
{
    const uint64_t groups = (nruns + 63) / 64;
    const uint64_t spread = groups < (uint64_t)num_cus ? groups : (uint64_t)num_cus;
    const uint64_t packed = (groups + waves - 1) / waves;
    uint64_t want = (2ull * num_cus + g_batch.count - 1) / g_batch.count;  // enough workgroups for two rounds of the chip
    if (want < packed) want = packed;
    return want < spread ? want : spread;
}

// SMARTGPU_DEBUG=1 in the environment: the geometry of every launch of a runs kernel on stderr (tools/nvar_probe.py)
static void trace_runs(const char* kernel, const ScanArgs& a, uint64_t run_len, const TileRange& tr, uint64_t grid)
{
    static const bool on = getenv("SMARTGPU_DEBUG") != nullptr;
    if (on)
        fprintf(stderr, "%s: starts [%llu, %llu) runs of %llu bytes: %llu from run %llu, %llu workgroups x %u patterns\n", kernel,
                (unsigned long long)a.s_begin, (unsigned long long)a.s_end, (unsigned long long)run_len,
                (unsigned long long)tr.count, (unsigned long long)tr.first, (unsigned long long)grid, g_batch.count);
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (device, kernel), not per launch
static void allow_lds(const void* kernel, size_t lds)
{
    static std::map<std::pair<int, const void*>, size_t> done;
    int dev = 0;
    (void)hipGetDevice(&dev);
    size_t& have = done[{dev, kernel}];
    if (have >= lds) return;
    (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    have = lds;
}

// Shift-Or runs (so_runs; Shift-And counts on it in complemented form); the masks u32 S[256] sit at
// a.blob + a.so_off.  shift_and / tune(6,4): the previous kernel so_runs1 (a step per byte, LineIo), A/B.
static hipError_t launch_so_runs(const ScanArgs& a, bool shift_and, int num_cus, hipStream_t stream, TextCodes codes)
{
    // bank-private table: 64 KB shared by the 16 waves of ONE workgroup per CU
    const uint32_t m = a.m;
    const uint64_t lmin = g_tune[5] ? std::min<uint64_t>((uint64_t)g_tune[5], kRunLenMax / 2) : 2048;
    const uint64_t L = balanced_run_len(a.s_begin, a.s_end, 64, (uint64_t)num_cus * kRunWaves, lmin, 2 * lmin, 128);
    const TileRange tr = tiles_for(a.s_begin, a.s_end, L);
    if (tr.count == 0) return hipSuccess;
    const bool four = codes.shift < 7 && g_tune[6] != 5;  // the text consists of at most four symbols
    const size_t lds = 65536 + kRunWaves * (size_t)kLineSlab + (four ? 1024 : 0);
    const uint64_t grid = runs_grid(tr.count, num_cus);
    trace_runs("so_runs", a, L, tr, grid);
#ifdef SMARTGPU_AB
#define SG_SO_RUNS1(L_, A_)                                                                               \
    do {                                                                                                 \
        allow_lds(reinterpret_cast<const void*>(so_runs1<L_, A_>), lds);                                  \
        hipLaunchKernelGGL((so_runs1<L_, A_>), dim3((uint32_t)grid, g_batch.count), dim3(64 * kRunWaves), lds, stream, a, \
                           (uint32_t)L, (uint64_t)tr.count, g_batch.items);                              \
    } while (0)
#endif
#define SG_SO_RUNS(L_, F_)                                                                               \
    do {                                                                                                 \
        allow_lds(reinterpret_cast<const void*>(so_runs<L_, F_>), lds);                                  \
        hipLaunchKernelGGL((so_runs<L_, F_>), dim3((uint32_t)grid, g_batch.count), dim3(64 * kRunWaves), lds, stream, a, \
                           (uint32_t)L, (uint64_t)tr.count, g_batch.items);                              \
    } while (0)
#ifdef SMARTGPU_AB
    if (shift_and) { if (m > 32) SG_SO_RUNS1(true, true); else SG_SO_RUNS1(false, true); }
    else if (g_tune[6] == 4) { if (m > 32) SG_SO_RUNS1(true, false); else SG_SO_RUNS1(false, false); }
    else
#undef SG_SO_RUNS1
#endif
    {
        if (four) { if (m > kSoWindow) SG_SO_RUNS(true, true); else SG_SO_RUNS(false, true); }
        else { if (m > kSoWindow) SG_SO_RUNS(true, false); else SG_SO_RUNS(false, false); }
    }
    (void)shift_and;
#undef SG_SO_RUNS
    return hipGetLastError();
}

static hipError_t launch_kmp_runs(const ScanArgs& a, int num_cus, hipStream_t stream, TextCodes codes)
{
    const uint32_t m = a.m;
    const uint32_t dfa_off = kTableOff + r16(2 * (m + 1));  // the transition table of kmp_runs (api.cpp build_blob)
    // four text bytes per table step: the text holds at most four byte values and the plan's window is short enough for
    // the table (a pattern over at most four symbols — any other cannot occur in such a text); tune(3,5): never (A/B)
    const bool four = a.prefer_packed != 0 && codes.shift < 7 && g_tune[3] != 5;
    uint32_t w = a.prefer_packed ? a.prefer_packed : kmp_window(m);  // bytes the automaton recognises (with that table: api.cpp); a run re-scans w-1
    const uint32_t rows = (w < 63 ? 4 * w + 2 : 256) + (four ? 2 : 0);  // up to the absorbing row Z (+ rows 4w+2, 4w+3 of the four-byte table)
#ifdef SMARTGPU_AB
    const bool links = g_tune[3] == 2;  // failure links
    const bool v1 = g_tune[3] == 3;     // the previous kernel (running maximum, half-line loader): its table follows
    const uint32_t dfa1_off = dfa_off + (w < 63 ? w + 1 : 256u) * 256 + kKmpQBytes;
    if (links || v1) w = (links || m <= kKmpDfaMaxM) ? m : kKmpDfaMaxM;
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
    const size_t table = v1 ? (size_t)(w < 64 ? 4 * w + 1 : 256) * 256 : (size_t)rows * 256;  // kmp_runs1: rows up to the accept id
#else
    const size_t table = (size_t)rows * 256;
#endif
    // one 1024-thread workgroup per CU shares the table (<= 64 KB) next to 16 x 4 KB of slabs
    const int waves = four ? kKmpFourWaves : kRunWaves;
    const size_t lds = table + kKmpQBytes + waves * (size_t)kLineSlab;
    // runs of 2-4 KiB: at least 8x the w-1 bytes a run re-scans, at most 8 KiB (the loader's
    // over-read past the last run stays inside the text's back pad)
    uint64_t lmin = g_tune[5] ? std::min<uint64_t>((uint64_t)g_tune[5], kRunLenMax / 2) : 2048;
    if (lmin < 8ull * (w - 1)) lmin = 8ull * (w - 1);
    const uint64_t lfloor = 2ull * (w - 1) > 128 ? 2ull * (w - 1) : 128;  // small texts: see balanced_run_len
    const uint64_t L = balanced_run_len(a.s_begin, a.s_end, 64, (uint64_t)num_cus * waves, lmin, 2 * lmin, lfloor);
    const TileRange tr = tiles_for(a.s_begin, a.s_end, L);
    if (tr.count == 0) return hipSuccess;
    const uint64_t grid = runs_grid(tr.count, num_cus, waves);
    trace_runs("kmp_runs", a, L, tr, grid);
#define SG_KMP_RUNS(K_, OFF_)                                                                            \
    do {                                                                                                 \
        if (lds > 64 * 1024) allow_lds(reinterpret_cast<const void*>(K_), lds);                          \
        hipLaunchKernelGGL(K_, dim3((uint32_t)grid, g_batch.count), dim3(64 * kRunWaves), lds, stream, a, (uint32_t)L, \
                           (uint64_t)tr.count, (uint32_t)(OFF_), g_batch.items);                         \
    } while (0)
#ifdef SMARTGPU_AB
    if (v1) {
        if (m > kKmpDfaMaxM) SG_KMP_RUNS(kmp_runs1<true>, dfa1_off); else SG_KMP_RUNS(kmp_runs1<false>, dfa1_off);
    } else
#endif
    {
        // tune(3,5): without the four-byte table — round 2's kernel on the same tables (A/B)
#define SG_KMP_RUNS4(P_, F_)                                                                             \
    do {                                                                                                 \
        if (lds > 64 * 1024) allow_lds(reinterpret_cast<const void*>(kmp_runs<P_, F_>), lds);            \
        hipLaunchKernelGGL((kmp_runs<P_, F_>), dim3((uint32_t)grid, g_batch.count), dim3(64 * waves), lds, stream, a, (uint32_t)L, \
                           (uint64_t)tr.count, dfa_off, g_batch.items);                                  \
    } while (0)
        if (four) {
            if (m > w) SG_KMP_RUNS4(true, true);  // beyond 62 bytes: the prefix's automaton
            else SG_KMP_RUNS4(false, true);
        } else {
            if (m > w) SG_KMP_RUNS4(true, false);  // beyond 254 bytes (62 with the four-byte table's window)
            else SG_KMP_RUNS4(false, false);
        }
#undef SG_KMP_RUNS4
    }
#undef SG_KMP_RUNS
    return hipGetLastError();
}

// The packed matcher; `a.blob` must carry the fingerprint at kTableOff (EPSM
// layout) — Horspool's blob carries it after its own tables (api.cpp).
template <int ALGO>
static hipError_t launch_packed(const ScanArgs& a, int num_cus, hipStream_t stream)
{
    const bool shuf = g_tune[7] == 3;
    const TileRange tr = tiles_for(a.s_begin, a.s_end, shuf ? (uint64_t)(kEpsmT / 64) * 1008 : (uint64_t)kEpsmT * 16);
    if (tr.count == 0) return hipSuccess;
    const int rows = 4;  // rows in flight per workgroup step (1 and 2 measured slower, profiles/r01)
    uint64_t grid = ((uint64_t)tr.count + rows - 1) / rows;
    // no LDS tile, no barrier in the loop: more, smaller shares balance better — 16 workgroups per
    // CU (two rounds) measured 76-79 % against 73-77 % with 8 on sparse hits and the same on dense
    // ones; beyond that the atomics on the result slot (one per workgroup with hits) show
    const uint64_t cap = (uint64_t)num_cus * (g_tune[4] ? g_tune[4] : 16);
    if (grid > cap) grid = cap;
#define SG_PACKED(M_, P_)                                                                           \
    hipLaunchKernelGGL((packed_scan<kEpsmT, 4, ALGO, M_, P_>), dim3((uint32_t)grid, g_batch.count), dim3(kEpsmT), 128, \
                       stream, a, tr.first, (uint64_t)tr.count, g_batch.items)
#ifdef SMARTGPU_AB  // the other load policies: both loads cached (1), one non-temporal load + shuffle (3)
#define SG_PACKED_POLICY(M_)                                                 \
    do {                                                                     \
        if (g_tune[7] == 1) SG_PACKED(M_, 1);                                \
        else if (g_tune[7] == 3) SG_PACKED(M_, 3);                           \
        else SG_PACKED(M_, 0);                                               \
    } while (0)
#else
#define SG_PACKED_POLICY(M_) SG_PACKED(M_, 0)
#endif
    if (a.m > 16) SG_PACKED_POLICY(2);
    else if (a.m % 4 == 0) SG_PACKED_POLICY(1);
    else SG_PACKED_POLICY(0);
#undef SG_PACKED_POLICY
#undef SG_PACKED
    return hipGetLastError();
}

// Regimes of the skip algorithms (HOR, BM, BNDM).  For short patterns the window is a
// few dwords and a skip loop degenerates (a lane advances ~m bytes per two dependent
// LDS reads); the packed matcher tests every alignment at HBM speed instead — the
// "hybrid" SURVEY.md §7 describes; counts are identical.  Measured on 1 GiB rand128
// (profiles/r01): HOR m=4: flat tile 45 %, bank-private 55 %, packed 74 % of 8 TB/s.
// Thresholds per algorithm: packed_max_m().

static int hor_regime(uint32_t m, int algo)
{
    const int v = g_tune[0];  // 0 auto, 1 flat, 2 bank-private, 3 packed
    if (v == 3) return 3;
    if (v == 2) return m <= kHaloMax + 1 ? 2 : 1;  // the bank-private kernel keeps whole windows in LDS
    if (v == 1) return 1;
    return m <= packed_max_m(algo) ? 3 : 1;  // bank-private kernel: only on request
}

// Positions through the packed matcher (any m); a.blob carries the fingerprint at a.fp_off,
// a.count receives the total number of occurrences.
hipError_t launch_find(const ScanArgs& a, unsigned long long* out, unsigned long long cap, int num_cus,
                       hipStream_t stream)
{
    if (a.s_end <= a.s_begin) return hipSuccess;
    const TileRange tr = tiles_for(a.s_begin, a.s_end, (uint64_t)kEpsmT * 16);
    if (tr.count == 0) return hipSuccess;
    uint64_t grid = tr.count;
    const uint64_t capg = (uint64_t)num_cus * 16;
    if (grid > capg) grid = capg;
    if (a.m > 16)
        hipLaunchKernelGGL((packed_find<kEpsmT, 2>), dim3((uint32_t)grid), dim3(kEpsmT), 0, stream, a, tr.first,
                           (uint64_t)tr.count, out, cap);
    else if (a.m % 4 == 0)
        hipLaunchKernelGGL((packed_find<kEpsmT, 1>), dim3((uint32_t)grid), dim3(kEpsmT), 0, stream, a, tr.first,
                           (uint64_t)tr.count, out, cap);
    else
        hipLaunchKernelGGL((packed_find<kEpsmT, 0>), dim3((uint32_t)grid), dim3(kEpsmT), 0, stream, a, tr.first,
                           (uint64_t)tr.count, out, cap);
    return hipGetLastError();
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
        case SMARTGPU_BNDML: a.fp_off = m > 32 ? kTableOff + 1024 * 2 + 4 : kTableOff + 1024; break;  // after the masks (W = 2) and the period / after B[256]
        case SMARTGPU_SBNDM:
        case SMARTGPU_BNDM: a.fp_off = kTableOff + 1024; break;  // after B[256]
        case SMARTGPU_EPSM: a.fp_off = kTableOff; break;
        case SMARTGPU_SO: a.so_off = kTableOff; break;
        case SMARTGPU_SA: a.so_off = g_tune[6] == 3 ? kTableOff + 1024 : kTableOff; break;
        case SMARTGPU_KMP: break;
        default: a.fp_off = kTableOff + 768; break;  // the Horspool family: after the u16 and u8 tables
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
    if (rerouted && g_tune[0] == 0) return launch_so_runs(a, false, num_cus, stream, codes);
    switch (algo) {
        case SMARTGPU_TUNEDBM:  // hor_scan<.., 0> is Tuned BM's loop (see the kernel's comment)
        case SMARTGPU_HOR: {
            const uint32_t H = a.halo;
            const int regime = (a.prefer_packed && g_tune[0] == 0) ? 3 : hor_regime(m);
            if (regime == 3) {
                return launch_packed<SMARTGPU_HOR>(a, num_cus, stream);  // a.fp_off: prepare_scan_args
            }
#ifdef SMARTGPU_AB
            if (regime == 2) {  // Horspool on the bank-private tiles (tune(0,2)); the product uses that kernel for Karp-Rabin only
                const size_t lds = kBpTabBytes + r16(H + 1) + kBpTextBytes;
                const TileRange tr = tiles_for(a.s_begin + m - 1, a.s_end + m - 1, (uint64_t)kBpThreads * kBpL);
                return launch_tiled(hor_scan_bp<false>, a, tr, kBpThreads, lds, 5, num_cus, stream);
            }
#endif
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
        case SMARTGPU_KR: {  // rolling hash on the bank-private tiles; a.halo = min(m-1, 32) (api.cpp)
            if (m < 16 && g_tune[0] != 1) {
                // Short patterns, like the skip algorithms': the packed matcher (72-77 %).  Only the low m bits
                // of the rolled hash can be compared, so one window end in 2^m is confirmed (m = 8: 32 %,
                // m = 12: 50 % of 8 TB/s), and below 8 the rolling form needs the outgoing byte (15 %).
                return launch_packed<SMARTGPU_HOR>(a, num_cus, stream);  // a.fp_off: prepare_scan_args
            }
            const uint32_t H = a.halo;
            const size_t lds = kBpTabBytes + r16(H + 1) + kBpTextBytes;
            const TileRange tr = tiles_for(a.s_begin + m - 1, a.s_end + m - 1, (uint64_t)kBpThreads * kBpL);
            return launch_tiled(hor_scan_bp<true>, a, tr, kBpThreads, lds, 6, num_cus, stream);
        }
        case SMARTGPU_HASH3:
        case SMARTGPU_HASH5:
        case SMARTGPU_HASH8:
        case SMARTGPU_RAITA:
        case SMARTGPU_QS: {  // the Horspool family on hor_scan's tiles; short patterns: packed regime as HOR
            const uint32_t H = a.halo;
            if ((a.prefer_packed && g_tune[0] == 0) || hor_regime(m, algo) == 3) {
                return launch_packed<SMARTGPU_HOR>(a, num_cus, stream);  // a.fp_off: prepare_scan_args
            }
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
#undef SG_HOR_VAR
            if (m - 1 > H) return launch_tiled(hor_scan<kHorT, kHorL, true, 1>, a, tr, kHorT, lds, tile_wgs(a), num_cus, stream);
            return launch_tiled(hor_scan<kHorT, kHorL, false, 1>, a, tr, kHorT, lds, tile_wgs(a), num_cus, stream);
        }
        case SMARTGPU_BM: {
            // m = 1 always: bm_scan reads the byte before the window's last along with it, and a one-byte
            // window has none (at a tile's first byte that read would leave the tile region)
            if (m == 1 || (m <= packed_max_m(SMARTGPU_BM) && g_tune[0] != 1) || (a.prefer_packed && g_tune[0] == 0)) {
                return launch_packed<SMARTGPU_BM>(a, num_cus, stream);  // a.fp_off: prepare_scan_args
            }
            const uint32_t H = a.halo;
            // Patterns whose symbols repeat (a.sparse == 0: natural language, small alphabets; only under tune(0,1) —
            // the plan sends them to so_runs): every lane is busy with candidates and a workgroup waits for its slowest
            // wave at each tile.  Two-wave workgroups, 12 per CU: English m = 4 / 8 / 32 / 128: 55 / 71 / 73 / 72 % against
            // 51 / 68 / 70 / 73 % with 6 four-wave workgroups (7: 52 / 63 / 69 / 70 %; 14 two-wave: 52 / 64 / 66 / 69 %).
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
        case SMARTGPU_BNDML:
            if (m > 32) {  // multi-word vectors; m <= 32 is plain BNDM (bndml.c:44-75): falls through
                const uint32_t w = m < kBndmlWindow ? m : kBndmlWindow;
                if (a.prefer_packed && g_tune[0] == 0) {
                    return launch_packed<SMARTGPU_BNDM>(a, num_cus, stream);  // a.fp_off: prepare_scan_args
                }
                const TileRange tr = tiles_for(a.s_begin + w - 1, a.s_end + w - 1, (uint64_t)kBndmT * kBndmL);
#define SG_BNDML(W_)                                                                                      \
    do {                                                                                                  \
        const size_t lds = 256 * (W_) * 4 + 256 + 256 + (size_t)kBndmT * kBndmL;                          \
        if (m > kBndmlWindow) return launch_tiled(bndml_scan<kBndmT, kBndmL, W_, true>, a, tr, kBndmT, lds, tile_wgs(a), num_cus, stream); \
        return launch_tiled(bndml_scan<kBndmT, kBndmL, W_, false>, a, tr, kBndmT, lds, tile_wgs(a), num_cus, stream);            \
    } while (0)
                static_assert(kBndmlWindow <= 64, "wider windows: instantiate bndml_scan with W = 4 (<= 128 bytes) or 8 (<= 256)");
                SG_BNDML(2);
#undef SG_BNDML
            }
            [[fallthrough]];
        case SMARTGPU_SBNDM:
        case SMARTGPU_BNDM: {
            if ((m <= packed_max_m(SMARTGPU_BNDM) && g_tune[0] != 1) || (a.prefer_packed && g_tune[0] == 0)) {
                return launch_packed<SMARTGPU_BNDM>(a, num_cus, stream);  // a.fp_off: prepare_scan_args
            }
            const uint32_t w = m < 32 ? m : 32;
            if (algo == SMARTGPU_SBNDM) {
                const size_t lds = 1024 + ((32 + (size_t)kBndmT * kBndmL + 63) & ~(size_t)63);  // whole 64-byte blocks (tile_at)
                const TileRange tr = tiles_for(a.s_begin + w - 1, a.s_end + w - 1, (uint64_t)kBndmT * kBndmL);
                if (m > 32) return launch_tiled(sbndm_scan<kBndmT, kBndmL, true>, a, tr, kBndmT, lds, tile_wgs(a), num_cus, stream);
                return launch_tiled(sbndm_scan<kBndmT, kBndmL, false>, a, tr, kBndmT, lds, tile_wgs(a), num_cus, stream);
            }
            // BNDM (and BNDML's m <= 32): q bytes of a window per iteration, q = a.halo from the plan (api.cpp build_blob).
            // Workgroups per CU (26.6 KB of LDS each, six fit), measured on 1 GiB (ms): a streaming scan (a.sparse,
            // rand128 m = 16 / 32 / 256) 0.170 / 0.160 / 0.173 with FOUR (five: 0.176 / 0.178 / 0.184); where windows
            // survive — English m = 16 / 32 / 256: 0.179 / 0.179 / 0.185 with FIVE (four: 0.179 / 0.174 / 0.204, six:
            // 0.183 / 0.182 / 0.192); a small alphabet (q >= 4), rand4 m = 16 / 32, rand2 m = 32: 0.185 / 0.172 / 0.195
            // with SIX (four: 0.202 / 0.165 / 0.214).  Two-wave workgroups (tune(2,2)) were never ahead: 8 of them
            // 0.190 / 0.171 / 0.200 on the small alphabets, 0.177 / 0.172 / 0.180 on English.
            uint32_t q = g_tune[1] ? (uint32_t)g_tune[1] : a.halo;  // tune(1, q): experiments
            while (q > 1 && w % q) q /= 2;
            const bool two_wave = g_tune[2] == 2;
            const int wgs = a.sparse ? 4 : q >= 4 ? 6 : 5;
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
        case SMARTGPU_SA:  // Shift-And: so_runs1<.., AND = true>; the A/B kernels below are Shift-Or only
        case SMARTGPU_SO: {
#ifdef SMARTGPU_AB
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
#endif
            // a.so_off: prepare_scan_args.  Shift-And counts in the complemented, Shift-Or form (api.cpp build_blob); its
            // own AND form (so_runs1<.., AND = true>, masks after the Shift-Or ones) is in the A/B build: tune(6,3)
            return launch_so_runs(a, algo == SMARTGPU_SA && g_tune[6] == 3, num_cus, stream, codes);
        }
        case SMARTGPU_KMP: {
#ifdef SMARTGPU_AB
            // tune(3,1), m <= 40: LDS tiles, run length per lane 80 or 144 bytes (re-scan of m-1 bytes <~28 %), failure links
            const size_t fixed = r16(4 * m) + r16(m - 1);
#define SG_KMP(T_, L_, WGS_)                                                                   \
    do {                                                                                       \
        const size_t lds = fixed + (size_t)(T_) * (L_);                                        \
        if (lds > 64 * 1024) allow_lds(reinterpret_cast<const void*>(kmp_scan<T_, L_>), lds);  \
        const TileRange tr = tiles_for(a.s_begin, a.s_end, (uint64_t)(T_) * (L_));            \
        return launch_tiled(kmp_scan<T_, L_>, a, tr, T_, lds, WGS_, num_cus, stream);          \
    } while (0)
            if (g_tune[3] == 1) {
                if (m <= 16) SG_KMP(256, 80, 6);
                if (m <= 40) SG_KMP(256, 144, 4);
            }
#undef SG_KMP
#endif
            // per-lane runs streamed through LDS
            return launch_kmp_runs(a, num_cus, stream, codes);
        }
        case SMARTGPU_EPSM: {
            return launch_packed<SMARTGPU_EPSM>(a, num_cus, stream);  // a.fp_off: prepare_scan_args
        }
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

hipError_t launch_generate(uint8_t* dst, uint64_t seed, int sigma, uint64_t off, uint64_t n,
                           hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    const uint64_t words = ((off + n + 7) >> 3) - (off >> 3);
    uint64_t grid = (words + 255) / 256;
    if (grid > 65536) grid = 65536;
    hipLaunchKernelGGL(generate_text, dim3((uint32_t)grid), dim3(256), 0, stream, dst, seed,
                       (uint32_t)sigma, off, n);
    return hipGetLastError();
}

hipError_t launch_tile_fill(uint8_t* dst, const uint8_t* unit, uint64_t unit_len, uint64_t phase,
                            uint64_t n, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    uint64_t grid = (n + 255) / 256;
    if (grid > 65536) grid = 65536;
    hipLaunchKernelGGL(tile_fill, dim3((uint32_t)grid), dim3(256), 0, stream, dst, unit, unit_len,
                       phase, n);
    return hipGetLastError();
}

}  // namespace sg
