/*
 * smart_oracle.c — CPU restatement of SMART's exact-matching hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see smart_oracle.h).  Parity status: PINNED
 * against the real reference builds in oracle/_ref/ and the golden vectors
 * in tests/golden/.
 *
 * Every function states the algorithm of the cited reference file in this
 * project's own words; none of it is copied.  Differences from the reference
 * that are deliberate (and covered by fixtures):
 *   - counts and text lengths are 64-bit;
 *   - nothing reads T[n..] (so.c:90 / bndm.c:101 do, SURVEY.md §5 hazard 1);
 *   - EPSM counts every occurrence (epsm.c:330 drops one at s=n-m when
 *     m%8==0, SURVEY.md §5 hazard 4) and works for n<16 (epsm.c:113 does not).
 *
 * Build: gcc -O3 -msse4.2 -fPIC -shared -pthread (oracle/Makefile).
 */
#include "smart_oracle.h"

#include <nmmintrin.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ */
/* Brute force — the reference's own ground truth.                     */
/* reference: src/algos/bf.c:25-39, src/test.c:45-56                   */
/* ------------------------------------------------------------------ */
uint64_t oracle_bf(const uint8_t *P, int m, const uint8_t *T, uint64_t n)
{
    if (m <= 0 || (uint64_t)m > n) return 0;
    uint64_t hits = 0;
    const uint64_t last = n - (uint64_t)m;
    for (uint64_t s = 0; s <= last; ++s) {
        int k = 0;
        while (k < m && P[k] == T[s + k]) ++k;
        hits += (k == m);
    }
    return hits;
}

/* ------------------------------------------------------------------ */
/* Horspool                                                            */
/* reference: Pre_Horspool src/algos/hor.c:26-30, search hor.c:33-51    */
/* ------------------------------------------------------------------ */
void oracle_pre_hor(const uint8_t *P, int m, int32_t hbc[ORACLE_SIGMA])
{
    /* shift = distance from the right-most occurrence of c in P[0..m-2]
     * to the last window position; m when c does not occur there. */
    for (int c = 0; c < ORACLE_SIGMA; ++c) hbc[c] = m;
    for (int i = 0; i + 1 < m; ++i) hbc[P[i]] = m - 1 - i;
}

uint64_t oracle_hor(const uint8_t *P, int m, const uint8_t *T, uint64_t n)
{
    if (m <= 0 || (uint64_t)m > n) return 0;
    int32_t hbc[ORACLE_SIGMA];
    oracle_pre_hor(P, m, hbc);
    uint64_t hits = 0, s = 0;
    const uint64_t last = n - (uint64_t)m;
    while (s <= last) {
        /* forward verification of the whole window (hor.c:44-46) */
        int k = 0;
        while (k < m && P[k] == T[s + k]) ++k;
        hits += (k == m);
        /* bad-character shift on the window's last byte (hor.c:47) */
        s += (uint64_t)hbc[T[s + m - 1]];
    }
    return hits;
}

/* ------------------------------------------------------------------ */
/* Boyer-Moore (bad character + good suffix)                           */
/* reference: preBmBc bm.c:27-33, suffixes bm.c:36-52,                  */
/*            preBmGs bm.c:54-66, search bm.c:69-93                     */
/* ------------------------------------------------------------------ */
void oracle_pre_bm_suffixes(const uint8_t *P, int m, int32_t *suff)
{
    /* suff[i] = length of the longest common suffix of P[0..i] and P.
     * Linear-time computation with the usual (f,g) window: g is the
     * left end (exclusive) of the right-most known suffix match that
     * started at f. */
    int f = 0, g = m - 1;
    suff[m - 1] = m;
    for (int i = m - 2; i >= 0; --i) {
        if (i > g && suff[i + (m - 1 - f)] < i - g) {
            suff[i] = suff[i + (m - 1 - f)];
        } else {
            if (i < g) g = i;
            f = i;
            while (g >= 0 && P[g] == P[g + (m - 1 - f)]) --g;
            suff[i] = f - g;
        }
    }
}

void oracle_pre_bm_gs(const uint8_t *P, int m, int32_t *gs)
{
    int32_t *suff = (int32_t *)malloc(sizeof(int32_t) * (size_t)(m > 0 ? m : 1));
    oracle_pre_bm_suffixes(P, m, suff);
    for (int i = 0; i < m; ++i) gs[i] = m;
    /* case 2: a prefix of P is a suffix of the matched part */
    int j = 0;
    for (int i = m - 1; i >= 0; --i) {
        if (suff[i] == i + 1) {
            for (; j < m - 1 - i; ++j)
                if (gs[j] == m) gs[j] = m - 1 - i;
        }
    }
    /* case 1: the matched suffix re-occurs inside P */
    for (int i = 0; i + 2 <= m; ++i) gs[m - 1 - suff[i]] = m - 1 - i;
    free(suff);
}

uint64_t oracle_bm(const uint8_t *P, int m, const uint8_t *T, uint64_t n)
{
    if (m <= 0 || (uint64_t)m > n) return 0;
    int32_t bc[ORACLE_SIGMA];
    int32_t *gs = (int32_t *)malloc(sizeof(int32_t) * (size_t)m);
    oracle_pre_bm_gs(P, m, gs);
    oracle_pre_hor(P, m, bc); /* bm.c:27-33 builds the same table as hor.c:26-30 */
    uint64_t hits = 0, s = 0;
    const uint64_t last = n - (uint64_t)m;
    while (s <= last) {
        int i = m - 1;
        while (i >= 0 && P[i] == T[s + (uint64_t)i]) --i;
        if (i < 0) {
            ++hits;
            s += (uint64_t)gs[0]; /* period of P (bm.c:86) */
        } else {
            int a = gs[i];
            int b = bc[T[s + (uint64_t)i]] - m + 1 + i;
            s += (uint64_t)(a > b ? a : b); /* bm.c:89 */
        }
    }
    free(gs);
    return hits;
}

/* ------------------------------------------------------------------ */
/* Knuth-Morris-Pratt (strong failure function)                        */
/* reference: preKmp kmp.c:27-41, search kmp.c:44-68                    */
/* ------------------------------------------------------------------ */
void oracle_pre_kmp(const uint8_t *P, int m, int32_t *next)
{
    int i = 0, j = -1;
    next[0] = -1;
    while (i < m) {
        while (j >= 0 && P[i] != P[j]) j = next[j];
        ++i;
        ++j;
        /* strong version: skip states that would fail on the same byte */
        next[i] = (i < m && P[i] == P[j]) ? next[j] : j;
    }
}

uint64_t oracle_kmp(const uint8_t *P, int m, const uint8_t *T, uint64_t n)
{
    if (m <= 0 || (uint64_t)m > n) return 0;
    int32_t *next = (int32_t *)malloc(sizeof(int32_t) * (size_t)(m + 1));
    oracle_pre_kmp(P, m, next);
    uint64_t hits = 0;
    int st = 0; /* number of pattern bytes currently matched */
    for (uint64_t j = 0; j < n; ++j) {
        while (st >= 0 && P[st] != T[j]) st = next[st];
        ++st;
        if (st >= m) {
            ++hits;
            st = next[st]; /* kmp.c:63 */
        }
    }
    free(next);
    return hits;
}

/* ------------------------------------------------------------------ */
/* Shift-Or, 32-bit words (WORD = 32, define.h:32)                     */
/* reference: preSo so.c:27-38, search so.c:40-61,                      */
/*            search_large so.c:69-96                                   */
/* ------------------------------------------------------------------ */
uint32_t oracle_pre_so(const uint8_t *P, int m, uint32_t S[ORACLE_SIGMA])
{
    /* S[c] has bit i CLEAR iff P[i]==c; returns `lim` such that
     * D < lim  <=>  bit m-1 of D is clear (all higher bits are set). */
    for (int c = 0; c < ORACLE_SIGMA; ++c) S[c] = 0xFFFFFFFFu;
    uint32_t used = 0;
    for (int i = 0; i < m; ++i) {
        S[P[i]] &= ~(1u << i);
        used |= 1u << i;
    }
    return ~(used >> 1);
}

uint64_t oracle_so(const uint8_t *P, int m, const uint8_t *T, uint64_t n)
{
    if (m <= 0 || (uint64_t)m > n) return 0;
    const int w = m > 32 ? 32 : m; /* so.c:44,73-74: prefix of 32 for long P */
    uint32_t S[ORACLE_SIGMA];
    const uint32_t lim = oracle_pre_so(P, w, S);
    uint64_t hits = 0;
    uint32_t D = 0xFFFFFFFFu;
    for (uint64_t j = 0; j < n; ++j) {
        D = (D << 1) | S[T[j]];
        if (D < lim) {
            if (w == m) {
                ++hits;
            } else {
                /* prefix hit at h; so.c:87-91 verifies all m bytes and may
                 * read past T[n-1] — here the window must lie inside T. */
                const uint64_t h = j + 1 - (uint64_t)w;
                if (h + (uint64_t)m <= n && memcmp(P + w, T + h + w, (size_t)(m - w)) == 0)
                    ++hits;
            }
        }
    }
    return hits;
}

/* ------------------------------------------------------------------ */
/* BNDM, 32-bit words                                                  */
/* reference: search bndm.c:27-62, search_large bndm.c:70-111           */
/* ------------------------------------------------------------------ */
void oracle_pre_bndm(const uint8_t *P, int m, uint32_t B[ORACLE_SIGMA])
{
    /* bit (m-1-i) of B[c] is set iff P[i]==c (bndm.c:35-40) */
    memset(B, 0, sizeof(uint32_t) * ORACLE_SIGMA);
    for (int i = 0; i < m; ++i) B[P[i]] |= 1u << (m - 1 - i);
}

uint64_t oracle_bndm(const uint8_t *P, int m, const uint8_t *T, uint64_t n)
{
    if (m <= 0 || (uint64_t)m > n) return 0;
    const int w = m > 32 ? 32 : m; /* bndm.c:31,74-75 */
    uint32_t B[ORACLE_SIGMA];
    oracle_pre_bndm(P, w, B);
    uint64_t hits = 0, s = 0;
    /* bndm.c:90 slides while the 32-byte prefix window fits; a prefix hit
     * whose full window does not fit is not an occurrence. */
    const uint64_t last = n - (uint64_t)w;
    while (s <= last) {
        int i = w - 1, shift = w;
        uint32_t D = 0xFFFFFFFFu;
        while (i >= 0 && D != 0) {
            D &= B[T[s + (uint64_t)i]];
            --i;
            if (D != 0) {
                if (i >= 0) {
                    shift = i + 1; /* a prefix of P ends here */
                } else if (w == m) {
                    ++hits;
                } else if (s + (uint64_t)m <= n &&
                           memcmp(P + w, T + s + w, (size_t)(m - w)) == 0) {
                    ++hits;
                }
            }
            D <<= 1;
        }
        s += (uint64_t)shift;
    }
    return hits;
}

/* ------------------------------------------------------------------ */
/* EPSM — exact packed string matching (SSE4.2)                        */
/* reference: src/algos/epsm.c  search1 :49, search2 :81, search3 :119, */
/*            search4 :165, search (5<=m<16) :341-419, search16 :225    */
/* Regimes follow the reference; block bookkeeping is this project's    */
/* own and never reads outside T[0..n).                                 */
/* ------------------------------------------------------------------ */
static inline uint32_t eq_mask16(__m128i blk, uint8_t c)
{
    return (uint32_t)_mm_movemask_epi8(_mm_cmpeq_epi8(blk, _mm_set1_epi8((char)c)));
}

/* m in {1,2,3}: broadcast compares, bit masks shifted into alignment with
 * a carry between 16-byte blocks (epsm.c:49-163). */
static uint64_t epsm_tiny(const uint8_t *P, int m, const uint8_t *T, uint64_t n)
{
    uint64_t hits = 0, b = 0;
    uint32_t carry0 = 0, carry1 = 0; /* bits of the previous block's masks */
    for (; b + 16 <= n; b += 16) {
        const __m128i blk = _mm_loadu_si128((const __m128i *)(T + b));
        const uint32_t e0 = eq_mask16(blk, P[0]);
        uint32_t endmask; /* bit p set: an occurrence ENDS at T[b+p] */
        if (m == 1) {
            endmask = e0;
        } else if (m == 2) {
            const uint32_t e1 = eq_mask16(blk, P[1]);
            endmask = ((e0 << 1) | carry0) & e1;
            carry0 = (e0 >> 15) & 1u;
        } else {
            const uint32_t e1 = eq_mask16(blk, P[1]);
            const uint32_t e2 = eq_mask16(blk, P[2]);
            endmask = ((e0 << 2) | carry0) & ((e1 << 1) | carry1) & e2;
            carry0 = (e0 >> 14) & 3u;
            carry1 = (e1 >> 15) & 1u;
        }
        hits += (uint64_t)_mm_popcnt_u32(endmask & 0xFFFFu);
    }
    /* remaining end positions, byte by byte (epsm.c:72-74,110-114,155-159) */
    for (uint64_t e = b; e < n; ++e) {
        if (e + 1 < (uint64_t)m) continue;
        hits += (memcmp(P, T + e + 1 - (uint64_t)m, (size_t)m) == 0);
    }
    return hits;
}

/* 16-bit mask of offsets o in [0,16) with T[o..o+4) == quad; reads t[0..24). */
static inline uint32_t quad_mask16(const uint8_t *t, __m128i quad)
{
    const __m128i lo = _mm_loadu_si128((const __m128i *)t);
    const __m128i hi = _mm_loadl_epi64((const __m128i *)(t + 16));
    const __m128i mid = _mm_alignr_epi8(hi, lo, 8); /* bytes 8..23 */
    const __m128i z = _mm_setzero_si128();
    /* mpsadbw: 8 sums of |t[o+i]-quad[i]|, i<4 — zero iff the 4 bytes match
     * (the filter epsm.c:195-205,379-381 is built on) */
    const __m128i s0 = _mm_cmpeq_epi16(_mm_mpsadbw_epu8(lo, quad, 0), z);
    const __m128i s1 = _mm_cmpeq_epi16(_mm_mpsadbw_epu8(mid, quad, 0), z);
    return (uint32_t)_mm_movemask_epi8(_mm_packs_epi16(s0, s1));
}

/* 4 <= m < 16 (epsm.c:165-223 for m==4, :341-419 for 5..15): 4-byte packed
 * filter at every alignment, memcmp on filter hits. */
static uint64_t epsm_short(const uint8_t *P, int m, const uint8_t *T, uint64_t n)
{
    /* filter bytes: whole pattern for m==4, P[m-5..m-2] otherwise (epsm.c:363) */
    const int foff = (m == 4) ? 0 : m - 5;
    uint32_t q32;
    memcpy(&q32, P + foff, 4);
    const __m128i quad = _mm_cvtsi32_si128((int)q32);
    const uint64_t last = n - (uint64_t)m; /* last valid start */
    uint64_t hits = 0, b = 0;            /* b = text offset of the filter window */
    for (; b + 24 <= n; b += 16) {
        uint32_t mask = quad_mask16(T + b, quad);
        if (m == 4) {
            hits += (uint64_t)_mm_popcnt_u32(mask);
            continue;
        }
        while (mask) {
            const uint64_t q = b + (uint64_t)__builtin_ctz(mask);
            mask &= mask - 1;
            if (q < (uint64_t)foff) continue;
            const uint64_t s = q - (uint64_t)foff;
            if (s <= last && memcmp(P, T + s, (size_t)m) == 0) ++hits;
        }
    }
    /* filter positions not covered by full 24-byte reads */
    for (uint64_t q = b; q + 4 <= n; ++q) {
        if (q < (uint64_t)foff) continue;
        const uint64_t s = q - (uint64_t)foff;
        if (s <= last && memcmp(P, T + s, (size_t)m) == 0) ++hits;
    }
    return hits;
}

/* m >= 16 (epsm.c:225-338): 11-bit crc32 fingerprints of every 8-byte
 * substring P[i..i+8), i in [1, 8*(m/8 - 1)], chained per bucket; the text
 * is probed at one aligned 8-byte word every 8*(m/8 - 1) bytes, so each
 * occurrence contains exactly one probe at one of those offsets. */
#define EPSM_SEED 123456789ULL /* epsm.c:231 */
#define EPSM_BUCKETS 2048      /* epsm.c:227 (HASHSIZE 11) */
static uint64_t epsm_long(const uint8_t *P, int m, const uint8_t *T, uint64_t n)
{
    const int stride = 8 * (m / 8 - 1); /* bytes between probes, >= 8 */
    int16_t head[EPSM_BUCKETS];
    int16_t *chain = (int16_t *)malloc(sizeof(int16_t) * (size_t)(stride + 1));
    memset(head, 0xFF, sizeof head);
    for (int i = stride; i >= 1; --i) { /* reverse so chains ascend like epsm.c:241-266 */
        uint64_t w;
        memcpy(&w, P + i, 8);
        const uint32_t h = (uint32_t)(_mm_crc32_u64(EPSM_SEED, w) & (EPSM_BUCKETS - 1));
        chain[i] = head[h];
        head[h] = (int16_t)i;
    }
    const uint64_t last = n - (uint64_t)m;
    uint64_t hits = 0;
    for (uint64_t q = (uint64_t)stride; q + 8 <= n; q += (uint64_t)stride) {
        uint64_t w;
        memcpy(&w, T + q, 8);
        const uint32_t h = (uint32_t)(_mm_crc32_u64(EPSM_SEED, w) & (EPSM_BUCKETS - 1));
        for (int i = head[h]; i >= 0; i = chain[i]) {
            const uint64_t s = q - (uint64_t)i;
            if (s <= last && memcmp(P, T + s, (size_t)m) == 0) ++hits;
        }
    }
    free(chain);
    return hits;
}

uint64_t oracle_epsm(const uint8_t *P, int m, const uint8_t *T, uint64_t n)
{
    if (m <= 0 || (uint64_t)m > n) return 0;
    if (m <= 3) return epsm_tiny(P, m, T, n);   /* epsm.c:343-345 */
    if (m < 16) return epsm_short(P, m, T, n);  /* epsm.c:346,348-418 */
    return epsm_long(P, m, T, n);               /* epsm.c:347 */
}

/* ------------------------------------------------------------------ */
/* Adjacent algorithms on the same engine.           */
/* ------------------------------------------------------------------ */

/* Shift-And: the dual of Shift-Or (a set bit = "prefix of that length
 * ends here").  reference: preSA src/algos/sa.c:27-34, search sa.c:36-56,
 * search_large (m > 32: 32-byte prefix + verification) sa.c:64-94.
 * Like oracle_so, the long-pattern path only accepts windows inside T. */
uint64_t oracle_sa(const uint8_t *P, int m, const uint8_t *T, uint64_t n)
{
    if (m <= 0 || (uint64_t)m > n) return 0;
    const int w = m < 32 ? m : 32;
    uint32_t S[ORACLE_SIGMA];
    memset(S, 0, sizeof S);
    for (int i = 0; i < w; ++i) S[P[i]] |= 1u << i;
    const uint32_t final = 1u << (w - 1);
    uint64_t hits = 0;
    uint32_t D = 0;
    for (uint64_t j = 0; j < n; ++j) {
        D = ((D << 1) | 1u) & S[T[j]];
        if (D & final) {
            const uint64_t s = j + 1 - (uint64_t)w;
            if (m == w) { ++hits; continue; }
            if (s + (uint64_t)m > n) continue;
            int k = w;
            while (k < m && P[k] == T[s + k]) ++k;
            hits += (k == m);
        }
    }
    return hits;
}

/* Quick Search: the shift is read from the byte AFTER the window.
 * reference: preQsBc src/algos/qs.c:27-31, search qs.c:33-52.  The
 * reference reads T[n] for the last window (its value no longer matters);
 * this restatement does not. */
uint64_t oracle_qs(const uint8_t *P, int m, const uint8_t *T, uint64_t n)
{
    if (m <= 0 || (uint64_t)m > n) return 0;
    int32_t shift[ORACLE_SIGMA];
    for (int c = 0; c < ORACLE_SIGMA; ++c) shift[c] = m + 1;
    for (int i = 0; i < m; ++i) shift[P[i]] = m - i;
    uint64_t hits = 0, s = 0;
    const uint64_t last = n - (uint64_t)m;
    while (s <= last) {
        int k = 0;
        while (k < m && P[k] == T[s + k]) ++k;
        hits += (k == m);
        if (s == last) break;
        s += (uint64_t)shift[T[s + m]];
    }
    return hits;
}

/* Tuned Boyer-Moore: Horspool's table with a zero for the pattern's last
 * byte drives a skip loop; a candidate is compared and the window then
 * moves by the shift the zero replaced.  reference: src/algos/tunedbm.c:27-65
 * (it plants m copies of P[m-1] after the text as a sentinel, :41; here the
 * skip loop checks the bound instead). */
uint64_t oracle_tunedbm(const uint8_t *P, int m, const uint8_t *T, uint64_t n)
{
    if (m <= 0 || (uint64_t)m > n) return 0;
    int32_t bc[ORACLE_SIGMA];
    oracle_pre_hor(P, m, bc);
    const int32_t after = bc[P[m - 1]];
    bc[P[m - 1]] = 0;
    uint64_t hits = 0, s = 0;
    const uint64_t last = n - (uint64_t)m;
    while (s <= last) {
        int32_t k;
        while ((k = bc[T[s + m - 1]]) != 0) {
            s += (uint64_t)k;
            if (s > last) return hits;
        }
        hits += (memcmp(P, T + s, (size_t)m - 1) == 0);
        s += (uint64_t)after;
    }
    return hits;
}

/* Raita: Horspool's shifts; the window is tested last byte, middle byte,
 * first byte, then the rest.  reference: src/algos/raita.c:27-64 (returns
 * -1 for m < 2, :37 — here: brute force, the count is defined anyway). */
uint64_t oracle_raita(const uint8_t *P, int m, const uint8_t *T, uint64_t n)
{
    if (m <= 0 || (uint64_t)m > n) return 0;
    if (m < 2) return oracle_bf(P, m, T, n);
    int32_t bc[ORACLE_SIGMA];
    oracle_pre_hor(P, m, bc);
    const uint8_t first = P[0], middle = P[m / 2], lastc = P[m - 1];
    uint64_t hits = 0, s = 0;
    const uint64_t last = n - (uint64_t)m;
    while (s <= last) {
        const uint8_t c = T[s + m - 1];
        if (c == lastc && T[s + m / 2] == middle && T[s] == first &&
            memcmp(P + 1, T + s + 1, (size_t)m - 2) == 0)
            ++hits;
        s += (uint64_t)bc[c];
    }
    return hits;
}

/* Lecroq's q-gram hashing (HASHq, q = 3, 5, 8): the shift is looked up under
 * an 8-bit hash of the window's last q bytes, h = sum y[i-k] * 2^k mod 256;
 * the hash of the pattern's last q-gram has shift 0 and stops the skip loop,
 * the window is compared, and moves on by the shift that zero replaced.
 * reference: src/algos/hash3.c:28-84, hash5.c, hash8.c (they return -1 for
 * m < q and plant a copy of the pattern after the text as a sentinel; here
 * the skip loop checks the bound).  For m == q the reference hashes one byte
 * past the pattern (hash3.c:50 with i = m); this restatement hashes the last
 * q-gram — the count is the same, every occurrence still stops the loop. */
static inline uint32_t qgram_hash(const uint8_t *end, int q)
{
    uint32_t h = 0;
    for (int k = q - 1; k >= 0; --k) h = (h << 1) + end[-k];
    return h & 0xFFu;
}

static uint64_t hashq(const uint8_t *P, int m, const uint8_t *T, uint64_t n, int q)
{
    if (m <= 0 || (uint64_t)m > n) return 0;
    if (m < q) return oracle_bf(P, m, T, n);  /* "not applicable" in the reference */
    int32_t shift[256];
    for (int i = 0; i < 256; ++i) shift[i] = m - q + 1;
    for (int i = q - 1; i < m - 1; ++i) shift[qgram_hash(P + i, q)] = m - 1 - i;
    const uint32_t hl = qgram_hash(P + m - 1, q);
    int32_t after = shift[hl];
    shift[hl] = 0;
    if (after == 0) after = 1;
    uint64_t hits = 0, e = (uint64_t)m - 1;  /* window end */
    while (e < n) {
        int32_t sh;
        while ((sh = shift[qgram_hash(T + e, q)]) != 0) {
            e += (uint64_t)sh;
            if (e >= n) return hits;
        }
        hits += (memcmp(P, T + e - (uint64_t)(m - 1), (size_t)m) == 0);
        e += (uint64_t)after;
    }
    return hits;
}

uint64_t oracle_hash3(const uint8_t *P, int m, const uint8_t *T, uint64_t n) { return hashq(P, m, T, n, 3); }
uint64_t oracle_hash5(const uint8_t *P, int m, const uint8_t *T, uint64_t n) { return hashq(P, m, T, n, 5); }
uint64_t oracle_hash8(const uint8_t *P, int m, const uint8_t *T, uint64_t n) { return hashq(P, m, T, n, 8); }

/* Simplified BNDM: the window is read right to left through the same
 * masks as BNDM, two bytes before the first test and no bookkeeping of
 * the longest prefix seen; a mismatch after k more bytes moves the window
 * past the failing byte, an occurrence moves it by the period of the
 * (32-byte prefix of the) pattern.  reference: src/algos/sbndm.c:28-83,
 * search_large (m > 32: prefix + verification, with a leading skip loop on
 * B[c] == 0) sbndm.c:91-149; -1 for m < 2 (:36) — here brute force. */
uint64_t oracle_sbndm(const uint8_t *P, int m, const uint8_t *T, uint64_t n)
{
    if (m <= 0 || (uint64_t)m > n) return 0;
    if (m < 2) return oracle_bf(P, m, T, n);
    const int w = m < 32 ? m : 32;
    uint32_t B[ORACLE_SIGMA];
    memset(B, 0, sizeof B);
    for (int p = 0; p < w; ++p) B[P[p]] |= 1u << (31 - p);
    /* period of P[0..w): w minus its longest proper border */
    int32_t border[33];
    border[0] = -1;
    for (int i = 0, b = -1; i < w; ++i) {
        while (b >= 0 && P[i] != P[b]) b = border[b];
        border[i + 1] = ++b;
    }
    const uint64_t period = (uint64_t)(w - border[w]);
    uint64_t hits = 0, e = (uint64_t)w - 1;  /* end of the w-byte window */
    const uint64_t e_last = n - (uint64_t)m + (uint64_t)w - 1;
    while (e <= e_last) {
        uint32_t D = B[T[e]];
        if (m > 32 && D == 0) { e += (uint64_t)w; continue; }  /* sbndm.c:133 */
        int k = 1;
        for (;;) {
            D = (D << 1) & B[T[e - (uint64_t)k]];
            if (k == w - 1 || D == 0) break;
            ++k;
        }
        if (D != 0) {
            const uint64_t s = e + 1 - (uint64_t)w;
            int q = w;
            while (q < m && P[q] == T[s + (uint64_t)q]) ++q;
            hits += (q == m);
            e += period;
        } else {
            e += (uint64_t)(w - k);
        }
    }
    return hits;
}

/* Karp-Rabin: a rolling hash of the window, h = sum T[s+i] * 2^(m-1-i) in
 * 32-bit arithmetic (so only the last 32 bytes of a longer window still
 * count), compared with the pattern's; equal hashes are confirmed byte by
 * byte.  reference: src/algos/kr.c:26-54 (REHASH :26; it reads T[n] when it
 * rolls past the last window, this restatement does not). */
uint64_t oracle_kr(const uint8_t *P, int m, const uint8_t *T, uint64_t n)
{
    if (m <= 0 || (uint64_t)m > n) return 0;
    uint32_t top = 1;  /* 2^(m-1) mod 2^32: the weight of the byte that leaves */
    for (int i = 1; i < m; ++i) top <<= 1;
    uint32_t hp = 0, ht = 0;
    for (int i = 0; i < m; ++i) {
        hp = (hp << 1) + P[i];
        ht = (ht << 1) + T[i];
    }
    uint64_t hits = 0;
    const uint64_t last = n - (uint64_t)m;
    for (uint64_t s = 0;; ++s) {
        if (hp == ht && memcmp(P, T + s, (size_t)m) == 0) ++hits;
        if (s == last) break;
        ht = ((ht - (uint32_t)T[s] * top) << 1) + T[s + (uint64_t)m];
    }
    return hits;
}

/* BNDM with multi-word bit vectors for patterns longer than a word: no
 * prefix trick, the whole pattern lives in ceil(m/32) words and the shift
 * carries from word to word.  reference: src/algos/bndml.c:44-75 (m <= 32:
 * plain BNDM), search_large bndml.c:82-132.  Bit i of B[c] <=> P[m-1-i] == c;
 * bit m-1 of D after k bytes <=> the last k window bytes are a prefix of P. */
uint64_t oracle_bndml(const uint8_t *P, int m, const uint8_t *T, uint64_t n)
{
    if (m <= 0 || (uint64_t)m > n) return 0;
    if (m <= 32) return oracle_bndm(P, m, T, n);
    const int W = (m + 31) / 32;
    uint32_t *B = calloc((size_t)ORACLE_SIGMA * (size_t)W, sizeof *B);
    uint32_t *D = calloc((size_t)W, sizeof *D);
    if (!B || !D) { free(B); free(D); return UINT64_MAX; }
    for (int i = 0; i < m; ++i) B[(size_t)P[m - 1 - i] * W + i / 32] |= 1u << (i % 32);
    uint64_t hits = 0, e = (uint64_t)m - 1;  /* window end */
    while (e < n) {
        const uint32_t *b = B + (size_t)T[e] * W;
        int alive = 0, k = 1, longest = 0;
        for (int i = 0; i < W; ++i) { D[i] = b[i]; alive |= D[i] != 0; }
        while (k < m && alive) {
            if (D[(m - 1) / 32] & (1u << ((m - 1) % 32))) longest = k;
            b = B + (size_t)T[e - (uint64_t)k] * W;
            uint32_t carry = 0;
            alive = 0;
            for (int i = 0; i < W; ++i) {
                const uint32_t cur = D[i];
                D[i] = ((cur << 1) | carry) & b[i];
                carry = cur >> 31;
                alive |= D[i] != 0;
            }
            ++k;
        }
        hits += alive != 0;
        e += (uint64_t)(m - longest);
    }
    free(B);
    free(D);
    return hits;
}

/* ------------------------------------------------------------------ */
/* dispatch                                                            */
/* ------------------------------------------------------------------ */
typedef uint64_t (*oracle_fn)(const uint8_t *, int, const uint8_t *, uint64_t);

static oracle_fn lookup(const char *name)
{
    static const struct { const char *name; oracle_fn fn; } tab[] = {
        {"bf", oracle_bf},     {"hor", oracle_hor},   {"bm", oracle_bm},
        {"kmp", oracle_kmp},   {"so", oracle_so},     {"bndm", oracle_bndm},
        {"epsm", oracle_epsm}, {"sa", oracle_sa},     {"qs", oracle_qs},
        {"tunedbm", oracle_tunedbm}, {"raita", oracle_raita},
        {"hash3", oracle_hash3}, {"hash5", oracle_hash5}, {"hash8", oracle_hash8},
        {"sbndm", oracle_sbndm}, {"kr", oracle_kr},       {"bndml", oracle_bndml},
    };
    for (size_t i = 0; i < sizeof tab / sizeof tab[0]; ++i)
        if (strcmp(tab[i].name, name) == 0) return tab[i].fn;
    return NULL;
}

uint64_t oracle_search(const char *name, const uint8_t *P, int m,
                       const uint8_t *T, uint64_t n)
{
    oracle_fn fn = lookup(name);
    return fn ? fn(P, m, T, n) : UINT64_MAX;
}

int oracle_search_int(const char *name, const uint8_t *P, int m,
                      const uint8_t *T, int n)
{
    if (n < 0) return -1;
    const uint64_t c = oracle_search(name, P, m, T, (uint64_t)n);
    return c > 0x7FFFFFFFull ? -1 : (int)c;
}

/* ------------------------------------------------------------------ */
/* multi-core driver (cpu_baseline only)                               */
/* ------------------------------------------------------------------ */
typedef struct {
    oracle_fn fn;
    const uint8_t *P, *T;
    int m;
    uint64_t n, hits;
} mt_job;

static void *mt_run(void *arg)
{
    mt_job *j = (mt_job *)arg;
    j->hits = j->fn(j->P, j->m, j->T, j->n);
    return NULL;
}

uint64_t oracle_search_mt(const char *name, const uint8_t *P, int m,
                          const uint8_t *T, uint64_t n, int threads)
{
    oracle_fn fn = lookup(name);
    if (!fn) return UINT64_MAX;
    if (m <= 0 || (uint64_t)m > n) return 0;
    if (threads < 1) threads = 1;
    const uint64_t starts = n - (uint64_t)m + 1;
    if ((uint64_t)threads > starts) threads = (int)starts;
    pthread_t *tid = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
    mt_job *job = (mt_job *)malloc(sizeof(mt_job) * (size_t)threads);
    for (int t = 0; t < threads; ++t) {
        /* thread t owns start positions [a,b); it sees bytes [a, b+m-1) */
        const uint64_t a = starts * (uint64_t)t / (uint64_t)threads;
        const uint64_t b = starts * (uint64_t)(t + 1) / (uint64_t)threads;
        job[t] = (mt_job){fn, P, T + a, m, (b - a) + (uint64_t)m - 1, 0};
        pthread_create(&tid[t], NULL, mt_run, &job[t]);
    }
    uint64_t hits = 0;
    for (int t = 0; t < threads; ++t) {
        pthread_join(tid[t], NULL);
        hits += job[t].hits;
    }
    free(job);
    free(tid);
    return hits;
}

/* ------------------------------------------------------------------ */
/* corpora                                                             */
/* ------------------------------------------------------------------ */
/* glibc random_r TYPE_3 (degree 31, separation 3) seeded with 1 — what an
 * unseeded rand() yields, which is what src/textgen.c:34-54 consumes. */
static void glibc_rand_init(uint32_t st[31], int *f, int *b)
{
    int32_t w = 1;
    st[0] = 1;
    for (int i = 1; i < 31; ++i) {
        /* w = 16807*w mod (2^31-1) by Schrage's method */
        const int32_t hi = w / 127773, lo = w % 127773;
        w = 16807 * lo - 2836 * hi;
        if (w < 0) w += 2147483647;
        st[i] = (uint32_t)w;
    }
    *f = 3;
    *b = 0;
}

static inline uint32_t glibc_rand_next(uint32_t st[31], int *f, int *b)
{
    st[*f] += st[*b];
    const uint32_t out = st[*f] >> 1;
    *f = (*f + 1) % 31;
    *b = (*b + 1) % 31;
    return out;
}

int oracle_textgen(int sigma, uint8_t *out, uint64_t n)
{
    static const int order[8] = {2, 4, 8, 16, 32, 64, 128, 250};
    int which = -1;
    for (int i = 0; i < 8; ++i)
        if (order[i] == sigma) which = i;
    if (which < 0 || n > 5000000ull) return -1;
    uint32_t st[31];
    int f, b;
    glibc_rand_init(st, &f, &b);
    for (int i = 0; i < 310; ++i) (void)glibc_rand_next(st, &f, &b);
    /* skip the corpora textgen writes before this one */
    for (uint64_t i = 0; i < 5000000ull * (uint64_t)which; ++i)
        (void)glibc_rand_next(st, &f, &b);
    for (uint64_t i = 0; i < n; ++i)
        out[i] = (uint8_t)(glibc_rand_next(st, &f, &b) % (uint32_t)sigma);
    return 0;
}

uint64_t oracle_splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

void oracle_gen_text(uint64_t seed, int sigma, uint64_t off, uint64_t n,
                     uint8_t *out)
{
    const int pow2 = (sigma & (sigma - 1)) == 0;
    for (uint64_t k = 0; k < n; ++k) {
        const uint64_t i = off + k;
        const uint64_t x = oracle_splitmix64(seed + (i >> 3));
        const uint32_t byte = (uint32_t)(x >> (8 * (i & 7))) & 0xFFu;
        out[k] = (uint8_t)(pow2 ? (byte & (uint32_t)(sigma - 1)) : (byte % (uint32_t)sigma));
    }
}
