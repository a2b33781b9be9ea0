// k_kmp.hip — KMP (the automaton of the failure function) over per-lane runs: kmp_runs
// (one translation unit per kernel family: dev_common.hpp)
#include "dev_common.hpp"
#include "runs_common.hpp"
#include "launch_common.hpp"

namespace sg {

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

// One transition.  C = false: the spread table of runs_common.hpp's kmp_delta (row id at id * 256).  C = true — the
// COMPACT table of kmp_runs<., false, true>: the rows of the states 0..w and Z one after the other, row s (id 4s) at
// s * 256 = id * 64, still XOR-swizzled by its id: delta(id, c) at (id << 6) | (c ^ id) (the ids are multiples of 4
// up to 252, so id << 6 has no bit below 8 that c ^ id could collide with).  v_xor_b32 (its byte operand selected
// from the text dword: SDWA) + v_lshl_or_b32 — two VALU operations on the chain, as v_perm_b32 + v_xor_b32 before.
template <bool C>
__device__ __forceinline__ uint32_t kmp_step(uint32_t dword, uint32_t st, int byte)
{
    if (!C) return kmp_delta(dword, st, byte);
    const uint32_t c = (dword >> (8 * byte)) & 0xFFu;
    const uint32_t addr = (st << 6) | (c ^ st);
    return *(const lds_u8_t*)(size_t)addr;
}

// sixteen transitions, nothing else (Z absorbs)
template <bool C>
__device__ __forceinline__ void kmp_chunk_fast(const uint4& v, uint32_t& st)
{
    const uint32_t d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int q = 0; q < 16; ++q) st = kmp_step<C>(d[q >> 2], st, q & 3);
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
template <bool EXT, bool C>
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
            for (int q = 0; q < 4; ++q) st = kmp_step<C>(d[k], st, q);
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
template <bool CHECK, bool MASK, bool C>
__device__ __forceinline__ void kmp_chunk_count(const uint4& v, uint32_t j_base, uint32_t j0, uint32_t jend,
                                                uint32_t& st, uint32_t& hits, uint32_t idw)
{
    const uint32_t d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const uint32_t nx = kmp_step<C>(d[q >> 2], st, q & 3);
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
// COMPACT (round 4) — FIVE four-wave workgroups per CU instead of one of sixteen waves.  The PMC passes of round 3 say
// the kernel waits (SQ_WAIT_ANY 65 % of the wave cycles; VALU, scalar unit and LDS each under a third busy): every wave
// has ONE line of its 64 runs in flight and one chain of dependent lookups, so what it needs is more waves, and the 94
// VGPRs of this kernel allow five per SIMD where a 1024-thread workgroup next to a spread table (33 KB at m = 32, 64 KB
// from 63 states on) allowed four.  So: the table COMPACT (kmp_step<true>: (w + 2) * 256 bytes, 16 KB at the most), the
// automaton that of the pattern or of its 56-byte prefix (kKmpCompactWindow — five times table + Q + four slabs must fit
// the CU's 160 KB of LDS), 256-thread workgroups — one wave per SIMD each, so any five co-reside.  The accept row's id
// must be a multiple of 4 like every other (row id * 64): Z = 4 (w + 1), and the counting walk's |next - min(next, 4w)|
// counts an occurrence as 4 (undone once, at the end).
// Measured against round 3's form (tune(3,6), A/B build, alternating, 1 GiB): m = 16: -6 % time on rand128, English and rand32;
// m = 32 ... 56: -3 ... -7 %; longer patterns (the prefix automaton) and rand256: equal.  Waves are what it lives on: the same
// kernel with four workgroups per CU +12 ... +16 %, and with 8 KB slabs — both halves of a line parked at once, the next line
// requested a half earlier, three workgroups per CU — +30 % (profiles/r04/e_ab_kmp_deep_*.log); six per CU do not fit its 96 VGPRs.
template <bool PREFIX, bool FOUR, bool COMPACT = false>  // PREFIX: the automaton of a prefix (62 bytes; COMPACT: 56); hits are verified
__global__ __launch_bounds__((COMPACT ? kKmpCompactWaves : FOUR ? kKmpFourWaves : kRunWaves) * 64, COMPACT ? kKmpCompactPerCu : FOUR ? 3 : 4)
void kmp_runs(ScanArgs a1, uint32_t run_len, uint64_t nruns, uint32_t dfa_off, const BatchItem* __restrict__ batch)
{
    static_assert(!(FOUR && COMPACT), "the four-byte table lives in the gaps of the spread table");
    const ScanArgs a = pick_args(a1, batch);  // a pattern set in one grid: blockIdx.y = pattern (launch_batch)
    constexpr int kW = COMPACT ? kKmpCompactWaves : FOUR ? kKmpFourWaves : kRunWaves;  // waves of the workgroup
    constexpr bool C = COMPACT;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t m = a.m;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t w = PREFIX ? (COMPACT ? kKmpCompactWindow : kKmpPrefix) : m;  // length the automaton recognises
    const uint32_t idw = (COMPACT || w < 63) ? 4 * w : 254u, Z = COMPACT ? idw + 4 : idw + 1;
    constexpr bool four = FOUR;
    // (+ two rows of that table: 4w + 2, the accept state's, and 4w + 3 = Z | 2, all Z: a lane that fell into Z stays there)
    const uint32_t table_bytes = COMPACT ? (w + 2) * 256 : (Z + 1 + (four ? 2u : 0u)) * 256;
    KmpPrefix4 pf;
    {
        const uint32_t p = *reinterpret_cast<const uint32_t*>(a.blob);  // P[0..4) (the pattern slot is zero-padded)
        pf.p4 = p;
        pf.p3 = p << 8;
        pf.p2 = p << 16;
        pf.p1 = p << 24;
    }
    const uint32_t qbase = table_bytes;  // Q[s] = P[s..s+4) for the states kmp_chunk_skip4 covers (256 bytes)
    const uint32_t stored = ((COMPACT || w < 63) ? w + 1 : 256u) * 256u;  // the blob's rows (tables.cpp kmp_runs_tables)
    const uint32_t thr = *reinterpret_cast<const uint32_t*>(a.blob + dfa_off + stored + 256);
    // FOUR: the text's two-bit codes, from the first words of the text's own allocation (TextCodes, kernels.hpp)
    const uint32_t shift4 = FOUR ? reinterpret_cast<const uint32_t*>(a.text - kFrontPad)[0] : 0u;
    const uint32_t four_symtab = FOUR ? reinterpret_cast<const uint32_t*>(a.text - kFrontPad)[1] : 0u;
    uint8_t* const slabs = smem + table_bytes + kKmpQBytes;
    const RunIo io = swap_io(slabs + wave * kLineSlab, lane, run_len);
    {
        const uint4* g = reinterpret_cast<const uint4*>(a.blob + dfa_off);
        uint4* t = reinterpret_cast<uint4*>(smem);
        if (COMPACT) {
            // the blob's rows as they are, row Z behind them, then Q
            for (uint32_t i = threadIdx.x; i < (w + 1) * 16; i += kW * 64) t[i] = g[i];
            const uint32_t z4 = Z * 0x01010101u;
            if (threadIdx.x < 16) t[(w + 1) * 16 + threadIdx.x] = make_uint4(z4, z4, z4, z4);
            else if (threadIdx.x < 32) t[table_bytes / 16 + threadIdx.x - 16] = g[stored / 16 + threadIdx.x - 16];
        } else if (w < 63) {
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
        // PREFIX: the first unverified prefix hit of this step, as the offset of its rest (text + start + w) in the lane's
        // run; 0: none (a rest starts at least w bytes into the run).  One register, not a flag and a pointer: the compact
        // instantiation has 96 to live in.
        uint32_t parked_off = 0;
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
                    if (whole) kmp_chunk_count<false, false, C>(v, j, j0, jend, st, hits, idw);
                    else kmp_chunk_count<true, false, C>(v, j, j0, jend, st, hits, idw);
                    return hits != h0;
                } else {
                    uint32_t hm = 0;
                    if (whole) kmp_chunk_count<false, true, C>(v, j, j0, jend, st, hm, idw);
                    else kmp_chunk_count<true, true, C>(v, j, j0, jend, st, hm, idw);
                    const bool seen = hm != 0;
                    while (hm) {  // the prefix ends at byte j+b: verify P[w..m)
                        const uint32_t b = __builtin_ctz(hm);
                        hm &= hm - 1;
                        const uint32_t off = j + b + 1;  // text + seg + off = text + start + w
                        if (parked_off == 0) parked_off = off;
                        else hits += global_equal(a.text + seg + off, a.blob + w, m - w);
                    }
                    return seen;
                }
            };
            if (jb >= j0 && jb + 64u <= jend) {  // the whole half is inside the run
                bool seen = false;
                if (!dense) {
                    // state before each 16-byte chunk.  The compact prefix automaton keeps only the half's first (a hit of a
                    // 56-byte prefix is rare, its half is walked again from the start): three registers of the 96 it has.
                    constexpr bool kOneSave = PREFIX && COMPACT;
                    uint32_t at[4];
                    if (mode == 1) {
                        bool low = __ballot(st != 0u) == 0;
                        uint32_t nfast = 0;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            at[q] = st;
                            kmp_chunk_skip4<false, C>(run_piece(io, q), st, low, pf, qbase, 0u, nfast);
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
                            kmp_chunk_skip4<true, C>(run_piece(io, q), st, low, pf, qbase, thr, nfast);
                        }
                        if (nfast < 6) mode = 0u;
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            at[q] = st;
                            kmp_chunk_fast<C>(run_piece(io, q), st);
                        }
                    }
                    seen = st == Z;
                    if (__any(seen)) {
                        if (seen) {
                            // the chunk in which the lane fell into Z: walk on from there, counting
                            const uint32_t q0 = kOneSave ? 0u : at[1] == Z ? 0u : at[2] == Z ? 1u : at[3] == Z ? 2u : 3u;
                            st = (kOneSave || q0 == 0) ? at[0] : q0 == 1 ? at[1] : q0 == 2 ? at[2] : at[3];
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
            if (PREFIX && __any(parked_off != 0)) {  // wave-uniform point: at most one parked hit per lane
                hits += wave_verify(parked_off != 0, a.text + seg + parked_off, a.blob + w, m - w);
                parked_off = 0;
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
    if (COMPACT && !PREFIX) hits >>= 2;  // the counting walk's |next - min(next, 4w)| is 4 per occurrence with Z = 4w + 4
    flush_hits(hits, a.count, smem, a.text);
}

// ---------------------------------------------------------------------------
// launcher (A/B build: tune(3, 1 / 2 / 3) select the superseded kernels — launch_ab_kmp, k_ab.hip — before this is called)
// ---------------------------------------------------------------------------
hipError_t launch_kmp_runs(const ScanArgs& a, int num_cus, hipStream_t stream, TextCodes codes)
{
    const uint32_t m = a.m;
    // four text bytes per table step: the text holds at most four byte values and the plan carries the spread table of a
    // window short enough for it (a.prefer_packed: a pattern over at most four symbols — any other cannot occur in
    // such a text); tune(3,5): never (A/B)
    const bool four = a.prefer_packed != 0 && codes.shift < 7 && g_tune[3] != 5;
#ifdef SMARTGPU_AB
    const bool spread = four || g_tune[3] == 6;  // tune(3,6): round 3's one-workgroup-per-CU form on the spread table (A/B)
#else
    const bool spread = four;
#endif
    uint64_t lmin = g_tune[5] ? std::min<uint64_t>((uint64_t)g_tune[5], kRunLenMax / 2) : 2048;
    if (!spread) {
        // Every other text: the compact table, five four-wave workgroups per CU.
        const uint32_t w = kmp_compact_window(m);  // bytes the automaton recognises; a run re-scans w-1
        const size_t lds = (size_t)(w + 2) * 256 + kKmpQBytes + kKmpCompactWaves * (size_t)kLineSlab;
        if (lmin < 8ull * (w - 1)) lmin = 8ull * (w - 1);
        const uint64_t lfloor = 2ull * (w - 1) > 128 ? 2ull * (w - 1) : 128;  // small texts: see balanced_run_len
        // workgroups per CU: five, if the runtime agrees that five are resident at once (a grid of five per CU of which four
        // fit runs a fifth of the text as a tail round: +25 %, measured with a 60-byte window); tune(4, .): A/B
        auto resident = [&](const void* kernel) {
            static std::map<std::pair<const void*, size_t>, int> known;
            int& n = known[{kernel, lds}];
            if (n == 0 && hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, 64 * kKmpCompactWaves, lds) != hipSuccess) n = kKmpCompactPerCu;
            return n < 1 ? 1 : n;
        };
        const void* const kern = m > w ? reinterpret_cast<const void*>(kmp_runs<true, false, true>) : reinterpret_cast<const void*>(kmp_runs<false, false, true>);
        const int per_cu = g_tune[4] ? g_tune[4] : std::min(kKmpCompactPerCu, resident(kern));
        const uint64_t L = balanced_run_len(a.s_begin, a.s_end, 64, (uint64_t)num_cus * per_cu * kKmpCompactWaves, lmin, 2 * lmin, lfloor);
        const TileRange tr = tiles_for(a.s_begin, a.s_end, L);
        if (tr.count == 0) return hipSuccess;
        const uint64_t grid = runs_grid(tr.count, num_cus, kKmpCompactWaves, per_cu);
        trace_runs(per_cu == kKmpCompactPerCu ? "kmp_runs (compact, 5 per CU)" : "kmp_runs (compact, FEWER than 5 per CU)", a, L, tr, grid);
        if (m > w)
            hipLaunchKernelGGL((kmp_runs<true, false, true>), dim3((uint32_t)grid, g_batch.count), dim3(64 * kKmpCompactWaves), lds, stream, a,
                               (uint32_t)L, (uint64_t)tr.count, kmp_compact_off(m), g_batch.items);
        else
            hipLaunchKernelGGL((kmp_runs<false, false, true>), dim3((uint32_t)grid, g_batch.count), dim3(64 * kKmpCompactWaves), lds, stream, a,
                               (uint32_t)L, (uint64_t)tr.count, kmp_compact_off(m), g_batch.items);
        return hipGetLastError();
    }
    const uint32_t dfa_off = kmp_spread_off(m);  // the spread table (api.cpp build_blob)
    const uint32_t w = a.prefer_packed ? a.prefer_packed : kmp_window(m);  // bytes the automaton recognises (with that table: api.cpp); a run re-scans w-1
    const uint32_t rows = (w < 63 ? 4 * w + 2 : 256) + (four ? 2 : 0);  // up to the absorbing row Z (+ rows 4w+2, 4w+3 of the four-byte table)
    const size_t table = (size_t)rows * 256;
    // one 1024-thread workgroup per CU shares the table (<= 64 KB) next to 16 x 4 KB of slabs
    const int waves = four ? kKmpFourWaves : kRunWaves;
    const size_t lds = table + kKmpQBytes + waves * (size_t)kLineSlab;
    // runs of 2-4 KiB: at least 8x the w-1 bytes a run re-scans, at most 8 KiB (the loader's
    // over-read past the last run stays inside the text's back pad)
    if (lmin < 8ull * (w - 1)) lmin = 8ull * (w - 1);
    const uint64_t lfloor = 2ull * (w - 1) > 128 ? 2ull * (w - 1) : 128;  // small texts: see balanced_run_len
    const uint64_t L = balanced_run_len(a.s_begin, a.s_end, 64, (uint64_t)num_cus * waves, lmin, 2 * lmin, lfloor);
    const TileRange tr = tiles_for(a.s_begin, a.s_end, L);
    if (tr.count == 0) return hipSuccess;
    const uint64_t grid = runs_grid(tr.count, num_cus, waves);
    trace_runs("kmp_runs", a, L, tr, grid);
#define SG_KMP_RUNS4(P_, F_)                                                                             \
    do {                                                                                                 \
        if (lds > 64 * 1024) allow_lds(reinterpret_cast<const void*>(kmp_runs<P_, F_>), lds);            \
        hipLaunchKernelGGL((kmp_runs<P_, F_>), dim3((uint32_t)grid, g_batch.count), dim3(64 * waves), lds, stream, a, (uint32_t)L, \
                           (uint64_t)tr.count, dfa_off, g_batch.items);                                  \
    } while (0)
    if (four) {
        if (m > w) SG_KMP_RUNS4(true, true);  // beyond 62 bytes: the prefix's automaton
        else SG_KMP_RUNS4(false, true);
    }
#ifdef SMARTGPU_AB
    else {
        if (m > w) SG_KMP_RUNS4(true, false);  // beyond 254 bytes (62 with the four-byte table's window)
        else SG_KMP_RUNS4(false, false);
    }
#endif
#undef SG_KMP_RUNS4
    return hipGetLastError();
}

}  // namespace sg
