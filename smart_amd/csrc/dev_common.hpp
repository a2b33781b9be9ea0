// dev_common.hpp — device-side helpers shared by every kernel translation unit (k_*.hip): the arguments of a
// workgroup, the non-temporal tile loads, hit reduction, wave-cooperative verification, the LDS tile layouts.
// Everything here is __forceinline__ device code; no kernel, no host state.
//
// One translation unit per kernel family (k_hor / k_bm / k_bndm / k_bndmx / k_so / k_kmp / k_packed / k_util, and
// k_ab in the A/B build): each is its own code object, so an edit to one kernel cannot move the code — and with it
// the instruction-cache alignment and the measured time — of another (round 3: a change in bm_scan moved so_runs by
// up to 5 % while all kernels shared kernels.hip).
#pragma once
#include "kernels.hpp"

#include <cstdio>
#include <cstdlib>
#include <type_traits>
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
// Round 4: a grid of a thousand workgroups and more that ALL have occurrences still ends in that many atomics on one
// address — 72 us behind a 0.67 ms scan of 4 GiB with packed_scan's 4096 workgroups (tools/hits_probe2.py: 44 us with
// 2048, nothing with 1024).  Such grids (one pattern per grid) add their sums to 64 staging slots first — 128 bytes apart
// in the text's own front pad (kHitSlotsOff: zero when a kernel starts, zero again when it ends; searches of one text
// are serialised on its device's stream) —, workgroup b to slot b % 64 together with a 1 in the slot's top 16 bits; the
// workgroup that finds all the others of its slot there adds the slot's sum to the result and clears it: 64 addresses with
// gridDim.x / 64 atomics each, then at most 64 on the result.
__device__ __forceinline__ void flush_hits(uint32_t lane_hits, unsigned long long* out, void* lds, const uint8_t* text)
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
        if (gridDim.y > 1 || gridDim.x < kHitSlotsMinGrid) {
            if (sum != 0) atomicAdd(out, sum);
        } else {
            const uint32_t g = blockIdx.x % kHitSlots;
            const uint32_t members = (gridDim.x - g + kHitSlots - 1) / kHitSlots;  // workgroups b = g (mod 64) of this grid
            unsigned long long* slot = reinterpret_cast<unsigned long long*>(const_cast<uint8_t*>(text) - kFrontPad + kHitSlotsOff) + 16u * g;
            const unsigned long long old = atomicAdd(slot, sum + (1ull << 48));   // (a sum stays below 2^48: at most one occurrence per text byte)
            if ((old >> 48) + 1 == members) {
                const unsigned long long total = (old + sum) & ((1ull << 48) - 1);
                atomicExch(slot, 0ull);
                if (total != 0) atomicAdd(out, total);
            }
        }
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

}  // namespace sg
