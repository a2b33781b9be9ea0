// k_packed.hip — the packed matcher: packed_scan (EPSM; the short-pattern regime of the skip algorithms), packed_find
// (one translation unit per kernel family: dev_common.hpp)
#include "dev_common.hpp"
#include "launch_common.hpp"

namespace sg {

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
struct EpsmFp { uint32_t f0, f1, f2, f3, k0, k1, k2, k3, nd, m; uint32_t q[16]; };  // q: modes 3 / 4

// text dword at byte offset x (compile-time after unrolling) of the 8-dword window d[]
#define SG_W(x) (((x) & 3) == 0 ? d[(x) >> 2] \
                                : __builtin_amdgcn_alignbyte(d[((x) >> 2) + 1], d[(x) >> 2], (x) & 3))

// MODE 0: some fingerprint dword is partial (m < 16, m % 4 != 0) -> masked compares;
// MODE 1: whole dwords only; MODE 2: m > 16, four whole dwords + bytes 16.. verified in memory
// MODE 3 / 4 (round 4) — a TEXT of TWO byte values (TextCodes.one: the bit that tells them apart), POLICY 6: a lane packs its
// own sixteen bytes into sixteen BITS (per dword: shift, and, one v_dot4 with the weights 1, 2, 4, 8) and takes the next lane's
// sixteen by ONE v_mov_b32_dpp wave_shl:1 (rows of 63 x 16 positions: the wave's last lane only supplies its bits).  The match
// vector of the sixteen alignments is computed bit-sliced over the PATTERN: M &= (W >> j) ^ (P[j] ? 0 : ~0) for j < F =
// min(m, 16) — a shift and one v_bitop3 per pattern symbol for all sixteen positions at once, (21 + 2 F) / 16 VALU ops per
// position where the byte-wise modes keep a candidate in every lane through all four fingerprint dwords (0.31-0.43 of the
// roofline from 16 bytes on).  3: m <= 16, exact; 4: m > 16, bytes 16.. of the survivors (one position in 65536) verified in
// memory.  fp.f0 = the pattern's F bits, fp.f1 = the text's bit, fp.k0 = F, fp.nd = 0 if a pattern byte is no symbol of the
// text (no occurrence: codes alias).  Own kernel, 1 GiB of two values: 9 … 4096 bytes 0.181-0.19 ms = 0.70-0.74 of the roofline
// (before: two bits per symbol, sixteen symbols per compare at each alignment: 0.56-0.58; four v_mqsad references: 0.55-0.58).
// MODES 5-10 (round 4) — epsm.c:165-223's mpsadbw filter with this machine's own instruction: v_mqsad_pk_u16_u8 takes eight
// text bytes and a four-byte reference and returns the sums of absolute differences at the FOUR byte alignments (16 bits
// each), skipping reference bytes that are 0, and ADDS them to its third operand.  Four pattern bytes are such a reference
// (the bytes beyond m are 0); a chain of up to four — the pattern's bytes 4s.. against the text 4s bytes on, accumulated —
// decides up to sixteen bytes: a sum of 0 is an occurrence.  Text and pattern are XORed with a byte K the pattern does not
// hold, so no pattern byte is 0.  A quarter-rate instruction (3.5 ordinary VALU slots each: tools/probe valu_rate; its
// semantics: tools/probe qsad_sem); the sums become flags with v_pk_min_u16(., 1) and are ADDED, two per dword: 16 - sum
// occurrences — no compare, no select, no ballot, whatever the text holds and however dense the occurrences are.
//   5 / 6 / 7 / 8: one .. four references, m <= 4 / 8 / 12 / 16, the whole pattern;
//   9 / 10: the first 16 / 8 bytes, the survivors' other bytes compared in memory (epsm_verify).
// fp.f0..f3 = the references, fp.k0 = K in every byte.
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
    if constexpr (MODE >= 5) {
        // references of four pattern bytes: modes 5-8: 1 (m <= 4) .. 4 (m <= 16), the whole pattern; modes 9 / 10: the first 16 / 8 bytes,
        // the survivors' other bytes compared in memory (epsm_verify)
        constexpr int NS = MODE == 9 ? 4 : MODE == 10 ? 2 : MODE - 4;
        constexpr bool SURVIVORS = MODE >= 9;
        constexpr int NX = 4 + NS;
        uint32_t x[NX];
#pragma unroll
        for (int j = 0; j < NX; ++j) x[j] = d[j] ^ fp.k0;
        uint32_t nz[8];  // nz[j]: bit 0 / bit 16 = position 2j / 2j + 1 is NO occurrence
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            // the pattern's bytes 4s .. 4s+3 against the text 4s bytes on, ADDED to the sums so far (the accumulator operand)
            uint64_t r = __builtin_amdgcn_mqsad_pk_u16_u8(((uint64_t)x[g4 + 1] << 32) | x[g4], fp.f0, 0ull);
            if (NS > 1) r = __builtin_amdgcn_mqsad_pk_u16_u8(((uint64_t)x[g4 + 2] << 32) | x[g4 + 1], fp.f1, r);
            if (NS > 2) r = __builtin_amdgcn_mqsad_pk_u16_u8(((uint64_t)x[g4 + 3] << 32) | x[g4 + 2], fp.f2, r);
            if (NS > 3) r = __builtin_amdgcn_mqsad_pk_u16_u8(((uint64_t)x[g4 + 4] << 32) | x[g4 + 3], fp.f3, r);
            // (as asm: from min(x, 1) on a vector of two the compiler makes two compares, two selects and a v_perm)
            asm("v_pk_min_u16 %0, %1, %2" : "=v"(nz[2 * g4]) : "v"((uint32_t)r), "v"(0x00010001u));
            asm("v_pk_min_u16 %0, %1, %2" : "=v"(nz[2 * g4 + 1]) : "v"((uint32_t)(r >> 32)), "v"(0x00010001u));
        }
        const uint32_t sum = (nz[0] + nz[1] + nz[2]) + (nz[3] + nz[4] + nz[5]) + (nz[6] + nz[7]);
        if (!SURVIVORS && !MASK && cand == 0xFFFFu)  // all sixteen positions are the lane's own: every row but the range's first and last
            return 16u - (sum & 0xFFFFu) - (sum >> 16);
        if (SURVIVORS && !MASK && !__any(sum != 0x00080008u)) return 0;  // no prefix in the wave's row: the streaming case
        uint32_t eq = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) eq |= (((nz[j] & 1u) ^ 1u) << (2 * j)) | ((((nz[j] >> 16) & 1u) ^ 1u) << (2 * j + 1));
        cand &= eq;
        if (SURVIVORS || MASK) {
            pending = cand;
            return 0;
        }
        return __popc(cand);
    }
    if constexpr (MODE == 3 || MODE == 4) {
        auto nib = [&](uint32_t x) -> uint32_t {  // four bytes -> four bits, the first in bit 0
            return __builtin_amdgcn_udot4((x >> fp.f1) & 0x01010101u, 0x08040201u, 0u, false);
        };
        const uint32_t w16 = nib(d[0]) | (nib(d[1]) << 4) | (nib(d[2]) << 8) | (nib(d[3]) << 12);
        const uint32_t W = w16 | (__builtin_amdgcn_update_dpp(0u, w16, 0x130, 0xF, 0xF, true) << 16);  // the next lane's sixteen bits above the lane's own
        // fp.q[j]: pattern symbol 1: 0 — the text's bits themselves; 0: ~0 — their complement (sixteen scalars, set once per kernel:
        // computed inside this loop they cost 4.5 scalar ops per symbol and the scalar unit, shared by the CU, bound the kernel)
        uint32_t M = fp.nd ? cand : 0u;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (MODE == 4 || (uint32_t)j < fp.k0) M &= (W >> j) ^ fp.q[j];  // (mode 4: F = 16; mode 3: a uniform test per symbol)
        }
        cand = M & 0xFFFFu;
        if (MODE == 4 || MASK) {
            pending = cand;
            return 0;
        }
        return __popc(cand);
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
                                                                 uint32_t m, uint32_t cand, uint64_t p0, uint32_t from = 16)
{
    // out of line on purpose: inlined, its control flow pushes the streaming loop of
    // packed_scan over the SGPR budget (spills into the hot path, -12 % measured)
    const uint32_t len = m - from;  // bytes from.. (16: the fingerprint's length; mode 10: 8)
    uint32_t c = cand;
    const bool has = c != 0;
    const uint32_t k0 = has ? __builtin_ctz(c) : 0u;
    c &= c - 1;
    cand &= ~(1u << k0);
    while (c) {
        const uint32_t k = __builtin_ctz(c);
        c &= c - 1;
        if (!global_equal(text + p0 + k + from, blob + from, len)) cand &= ~(1u << k);
    }
    return __popc(cand) + wave_verify(has, text + p0 + k0 + from, blob + from, len);
}

// ALGO only tags the instantiation (rocprofv3 shows packed_scan<256, 4, 5, ..> for
// EPSM and packed_scan<256, 4, 0, ..> for Horspool's short-pattern regime).
template <int THREADS, int ROWS, int ALGO, int MODE, int POLICY>
__global__ __launch_bounds__(THREADS) void packed_scan(ScanArgs a1, uint64_t row_first,
                                                       uint64_t nrows, const BatchItem* __restrict__ batch)
{
    const ScanArgs a = pick_args(a1, batch);  // a pattern set in one grid: blockIdx.y = pattern (launch_batch)
    // POLICY 0: A non-temporal, B cached (default); 1: both cached; 3: one nt load + shuffle; 5: one nt load + ONE dword by DPP (MODE 5).
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
    if constexpr (MODE >= 5) {  // the references of v_mqsad: P[0..min(m, 16)) ^ K, K = the smallest byte value those bytes do not hold
        constexpr uint32_t kF = MODE == 10 ? 8u : 16u;
        const uint32_t F = a.m < kF ? a.m : kF;
        uint32_t K = 0;
        for (uint32_t t = 0; t < 17; ++t) {
            bool held = false;
            for (uint32_t i = 0; i < F; ++i) held = held || a.blob[i] == K;
            if (!held) break;
            ++K;
        }
        uint32_t ref[4] = {0u, 0u, 0u, 0u};
        for (uint32_t i = 0; i < F; ++i) ref[i >> 2] |= ((uint32_t)a.blob[i] ^ K) << (8u * (i & 3u));
        fp.f0 = ref[0];
        fp.f1 = ref[1];
        fp.f2 = ref[2];
        fp.f3 = ref[3];
        fp.k0 = K * 0x01010101u;
    }
    if constexpr (MODE == 3 || MODE == 4) {  // the pattern's first F symbols as bits, under the TEXT's one-bit code (the first words of its allocation)
        const uint32_t one = reinterpret_cast<const uint32_t*>(a.text - kFrontPad)[2];
        const uint32_t bit = one & 0xFFu, s0 = (one >> 8) & 0xFFu, s1 = (one >> 16) & 0xFFu;
        const uint32_t F = a.m < 16 ? a.m : 16u;
        uint32_t pb = 0, ok = 1;
        for (uint32_t i = 0; i < F; ++i) {
            const uint32_t c = a.blob[i], code = (c >> bit) & 1u;
            pb |= code << i;
            ok &= (code ? s1 : s0) == c ? 1u : 0u;  // a byte the text does not hold aliases one it does
        }
        fp.f0 = pb;
        fp.f1 = bit;
        fp.k0 = F;
        fp.nd = ok;
#pragma unroll
        for (int j = 0; j < 16; ++j) fp.q[j] = 0u - ((~pb >> j) & 1u);
    }

    uint32_t hits = 0;
    // POLICY 3: a wave-row is 63*16 = 1008 start positions; lane i loads the 16 bytes at
    // row + 16*i ONCE (non-temporal) and takes the next 16 bytes from lane i+1 by a
    // cross-lane shuffle; lane 63 only supplies the overlap into the next wave-row.
    // Other policies: a row is THREADS*16 offsets and every lane loads 32 bytes.
    constexpr bool SHUF = POLICY == 3 || POLICY == 6;  // 6: modes 3 / 4 (the neighbour's BITS travel, inside epsm_row)
    constexpr uint32_t ROW_BYTES = SHUF ? (THREADS / 64) * 1008u : THREADS * 16u;
    const uint32_t in_row = SHUF ? (threadIdx.x >> 6) * 1008u + (threadIdx.x & 63u) * 16u : threadIdx.x * 16u;
    // a workgroup takes ROWS consecutive rows per step
    for (uint64_t g = (uint64_t)blockIdx.x * ROWS; g < nrows; g += (uint64_t)gridDim.x * ROWS) {
        uint4 A[ROWS], B[ROWS];
#pragma unroll
        for (int j = 0; j < ROWS; ++j) {
            const uint64_t r = g + j < nrows ? g + j : nrows - 1;  // clamp, ignored below
            const uint8_t* src = a.text + (row_first + r) * ROW_BYTES + in_row;
            if (POLICY == 5) {  // MODE 5 reads 19 bytes per lane: its own sixteen and ONE dword of the next lane's
                A[j] = ld_stream16(src);
                B[j] = uint4{0u, 0u, 0u, 0u};
                if ((threadIdx.x & 63u) == 63u) {  // the wave's last lane has no neighbour
                    if (MODE == 5) B[j].x = *reinterpret_cast<const uint32_t*>(src + 16);
                    else if (MODE == 6 || MODE == 10) { const uint2 t2 = *reinterpret_cast<const uint2*>(src + 16); B[j].x = t2.x; B[j].y = t2.y; }
                    else B[j] = *reinterpret_cast<const uint4*>(src + 16);  // (MODE 7: three dwords would do)
                }
            } else if (SHUF) {
                A[j] = ld_stream16(src);
                if (POLICY == 6) B[j] = uint4{0u, 0u, 0u, 0u};
            } else {
                // A is this lane's own 16 bytes; B re-reads the next lane's 16 bytes
                A[j] = NTA ? ld_stream16(src) : *reinterpret_cast<const uint4*>(src);
                B[j] = NTB ? ld_stream16(src + 16) : *reinterpret_cast<const uint4*>(src + 16);
            }
        }
        if (POLICY == 5) {
#pragma unroll
            for (int j = 0; j < ROWS; ++j) {  // v_mov_b32_dpp wave_shl:1 — lane i takes lane i + 1's dword, one VALU op, no second load
                const bool last = (threadIdx.x & 63u) == 63u;
                const uint32_t next = __builtin_amdgcn_update_dpp(0u, A[j].x, 0x130, 0xF, 0xF, true);
                B[j].x = last ? B[j].x : next;
                if (MODE != 5) {
                    const uint32_t ny = __builtin_amdgcn_update_dpp(0u, A[j].y, 0x130, 0xF, 0xF, true);
                    B[j].y = last ? B[j].y : ny;
                }
                if (MODE != 5 && MODE != 6 && MODE != 10) {
                    const uint32_t nz = __builtin_amdgcn_update_dpp(0u, A[j].z, 0x130, 0xF, 0xF, true);
                    const uint32_t nw = __builtin_amdgcn_update_dpp(0u, A[j].w, 0x130, 0xF, 0xF, true);
                    B[j].z = last ? B[j].z : nz;
                    B[j].w = last ? B[j].w : nw;
                }
            }
        }
        if (POLICY == 3) {
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
        if (MODE == 2 || MODE == 4 || MODE >= 9) {
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
                        hits += epsm_verify(a.text, a.blob, a.m, c, (row_first + g + j) * ROW_BYTES + in_row, MODE == 10 ? 8u : 16u);
                }
            }
        }
    }
    flush_hits(hits, a.count, smem, a.text);
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
// launchers.  The packed matcher; `a.blob` must carry the fingerprint at a.fp_off (EPSM layout) — the skip
// algorithms' blobs carry it after their own tables (api.cpp, prepare_scan_args).
// ---------------------------------------------------------------------------
template <int ALGO>
static hipError_t launch_packed_as(const ScanArgs& a, int num_cus, hipStream_t stream, TextCodes codes)
{
    const bool two_l = (codes.one & 0xFFu) != 0xFFu && codes.shift < 7;
    const bool bits = ALGO == SMARTGPU_EPSM && two_l && a.m >= 9 && (g_tune[7] == 0 || g_tune[7] == 5);  // modes 3 / 4 (below)
    const bool shuf = g_tune[7] == 3 || bits;
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
    // D_: the mode's own data path — 5 (one load, the neighbour's bytes by DPP) except for the masked compares of MODE 0
#ifdef SMARTGPU_AB  // the other load policies: both loads cached (1), one non-temporal load + shuffle (3)
#define SG_PACKED_POLICY(M_, D_)                                             \
    do {                                                                     \
        if (g_tune[7] == 1) SG_PACKED(M_, 1);                                \
        else if (g_tune[7] == 3) SG_PACKED(M_, 3);                           \
        else if (g_tune[7] == 7) SG_PACKED(M_, 0);                           \
        else SG_PACKED(M_, D_);                                              \
    } while (0)
#else
#define SG_PACKED_POLICY(M_, D_)                                             \
    do {                                                                     \
        if (g_tune[7] == 7) SG_PACKED(M_, 0);                                \
        else SG_PACKED(M_, D_);                                              \
    } while (0)
#endif
    // v_mqsad_pk_u16_u8 (modes 5-10): the same work whatever the text holds.  Up to 7 bytes always (0.78-0.82 against 0.69-0.76 for
    // the dword compares on rand128 / English, 0.50 on four symbols); 8+ bytes on a text of at most four byte values, where the
    // dword compares keep candidates alive through every stage (0.31-0.67); on other texts the dword compares' first stage
    // decides (8 bytes on rand128: 0.82 against 0.77).  ms per GiB, own kernel, 1 GiB:
    //   one / two references (m <= 8): 0.166-0.175 on any text;
    //   three or four values, 9+ bytes: two references — one position in 65536 survives — and the survivors completed in memory
    //     (mode 10): 0.170-0.173 up to 256 bytes (four references: 0.24-0.25; dword compares 0.21-0.23);
    //   two values, 9+ bytes: EPSM's bit-sliced modes 3 / 4: 0.181-0.19 (three references 0.19-0.20, four 0.23-0.24; three +
    //     survivors — one position in 4096 — 0.25-0.39; dword compares 0.32-0.44).
    // tune(7, 8) / (7, 9): the dword compares; tune(7, 6): references for every byte up to 16 (A/B)
    const bool few = codes.shift < 7, two = (codes.one & 0xFFu) != 0xFFu;
    const bool sad = g_tune[7] == 6 || (g_tune[7] == 0 && (a.m <= 7 || few));
    const bool symbols = bits;
    if constexpr (ALGO == SMARTGPU_EPSM) {
        if (symbols) {
            if (a.m > 16) SG_PACKED(4, 6);
            else SG_PACKED(3, 6);
        }
    }
    if (symbols) {}
    else if (sad && a.m <= 4) SG_PACKED(5, 5);
    else if (sad && a.m <= 8) SG_PACKED(6, 5);
    else if (sad && g_tune[7] != 6 && !two) SG_PACKED(10, 5);
    else if (sad && a.m <= 12) SG_PACKED(7, 5);
    else if (sad && a.m <= 16) SG_PACKED(8, 5);
    else if (sad) SG_PACKED(9, 5);
    else if (a.m <= 4 && g_tune[7] == 7) SG_PACKED(5, 0);  // (A/B: one stage with the second load)
    else if (a.m > 16) SG_PACKED_POLICY(2, 5);
    else if (a.m % 4 == 0) SG_PACKED_POLICY(1, 5);
    else SG_PACKED_POLICY(0, 0);  // masked compares: more VALU work per position, and the four DPP moves + selects cost it 2-3 points (measured)
#undef SG_PACKED_POLICY
#undef SG_PACKED
    return hipGetLastError();
}

hipError_t launch_packed(int kind, const ScanArgs& a, int num_cus, hipStream_t stream, TextCodes codes)
{
    switch (kind) {
        case SMARTGPU_HOR: return launch_packed_as<SMARTGPU_HOR>(a, num_cus, stream, codes);
        case SMARTGPU_BM: return launch_packed_as<SMARTGPU_BM>(a, num_cus, stream, codes);
        case SMARTGPU_BNDM: return launch_packed_as<SMARTGPU_BNDM>(a, num_cus, stream, codes);
        case SMARTGPU_EPSM: return launch_packed_as<SMARTGPU_EPSM>(a, num_cus, stream, codes);
    }
    return hipErrorInvalidValue;
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


}  // namespace sg
