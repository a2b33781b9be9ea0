// k_so.hip — Shift-Or / Shift-And over per-lane runs: so_runs
// (one translation unit per kernel family: dev_common.hpp)
#include "dev_common.hpp"
#include "runs_common.hpp"
#include "launch_common.hpp"

namespace sg {

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
    flush_hits(hits, a.count, smem, a.text);
}

// ---------------------------------------------------------------------------
// launcher: Shift-Or runs (Shift-And counts on it in complemented form); the masks u32 S[256] sit at a.blob + a.so_off.
// (A/B build: shift_and / tune(6,4) select the previous kernel so_runs1 — launch_ab_so, k_ab.hip — before this is called.)
// ---------------------------------------------------------------------------
hipError_t launch_so_runs(const ScanArgs& a, bool shift_and, int num_cus, hipStream_t stream, TextCodes codes)
{
    (void)shift_and;
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
#define SG_SO_RUNS(L_, F_)                                                                               \
    do {                                                                                                 \
        allow_lds(reinterpret_cast<const void*>(so_runs<L_, F_>), lds);                                  \
        hipLaunchKernelGGL((so_runs<L_, F_>), dim3((uint32_t)grid, g_batch.count), dim3(64 * kRunWaves), lds, stream, a, \
                           (uint32_t)L, (uint64_t)tr.count, g_batch.items);                              \
    } while (0)
    if (four) { if (m > kSoWindow) SG_SO_RUNS(true, true); else SG_SO_RUNS(false, true); }
    else { if (m > kSoWindow) SG_SO_RUNS(true, false); else SG_SO_RUNS(false, false); }
#undef SG_SO_RUNS
    return hipGetLastError();
}


}  // namespace sg
