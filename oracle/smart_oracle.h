/*
 * smart_oracle.h — CPU restatement of SMART's exact-matching hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / reported CPU baseline.  The
 * GPU library (smart_amd/csrc) never links or calls it.
 *
 * Parity status: PINNED.  Every function here is diffed against the real
 * reference algorithms (oracle/_ref/lib<algo>.so, built by oracle/Makefile
 * straight from /root/reference/src/algos/<algo>.c) by tests/test_oracle.py
 * when /root/reference is present, and against the committed golden vectors
 * (the JSON files in tests/golden/, generated from those same reference builds by
 * tests/golden/gen_golden.py) everywhere else.
 *
 * All functions count occurrences of P[0..m) in T[0..n), overlaps included
 * (reference: OUTPUT(j) == count++, src/algos/include/define.h:33), and read
 * only T[0..n).  Counts are 64-bit; the reference's `int search()` shape is
 * provided by oracle_search_int().
 */
#ifndef SMART_ORACLE_H
#define SMART_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORACLE_SIGMA 256
#define ORACLE_XSIZE 4200 /* src/algos/include/define.h:25 */

/* --- searches (reference file:line in smart_oracle.c above each body) --- */
uint64_t oracle_bf(const uint8_t *P, int m, const uint8_t *T, uint64_t n);
uint64_t oracle_hor(const uint8_t *P, int m, const uint8_t *T, uint64_t n);
uint64_t oracle_bm(const uint8_t *P, int m, const uint8_t *T, uint64_t n);
uint64_t oracle_kmp(const uint8_t *P, int m, const uint8_t *T, uint64_t n);
uint64_t oracle_so(const uint8_t *P, int m, const uint8_t *T, uint64_t n);
uint64_t oracle_bndm(const uint8_t *P, int m, const uint8_t *T, uint64_t n);
uint64_t oracle_epsm(const uint8_t *P, int m, const uint8_t *T, uint64_t n);
/* adjacent algorithms: sa.c, qs.c, tunedbm.c, raita.c */
uint64_t oracle_sa(const uint8_t *P, int m, const uint8_t *T, uint64_t n);
uint64_t oracle_qs(const uint8_t *P, int m, const uint8_t *T, uint64_t n);
uint64_t oracle_tunedbm(const uint8_t *P, int m, const uint8_t *T, uint64_t n);
uint64_t oracle_raita(const uint8_t *P, int m, const uint8_t *T, uint64_t n);
/* hash3.c, hash5.c, hash8.c (Lecroq's q-gram hashing) */
uint64_t oracle_hash3(const uint8_t *P, int m, const uint8_t *T, uint64_t n);
uint64_t oracle_hash5(const uint8_t *P, int m, const uint8_t *T, uint64_t n);
uint64_t oracle_hash8(const uint8_t *P, int m, const uint8_t *T, uint64_t n);
/* sbndm.c (simplified BNDM) */
uint64_t oracle_sbndm(const uint8_t *P, int m, const uint8_t *T, uint64_t n);
/* kr.c (Karp-Rabin) */
uint64_t oracle_kr(const uint8_t *P, int m, const uint8_t *T, uint64_t n);
/* bndml.c (BNDM with multi-word bit vectors) */
uint64_t oracle_bndml(const uint8_t *P, int m, const uint8_t *T, uint64_t n);

/* name in {"bf","hor","bm","kmp","so","bndm","epsm","sa","qs","tunedbm","raita","hash3","hash5","hash8","sbndm","kr","bndml"}; returns -1 for an
 * unknown name or a count that does not fit an int (SMART's -1 convention,
 * src/algos/include/main.h:39 and e.g. ssef.c:41). */
int oracle_search_int(const char *name, const uint8_t *P, int m,
                      const uint8_t *T, int n);
/* 64-bit dispatcher; returns UINT64_MAX for an unknown name. */
uint64_t oracle_search(const char *name, const uint8_t *P, int m,
                       const uint8_t *T, uint64_t n);

/* --- preprocessing tables (so tests can diff the GPU host-side builders) --- */
void oracle_pre_hor(const uint8_t *P, int m, int32_t hbc[ORACLE_SIGMA]);
void oracle_pre_bm_suffixes(const uint8_t *P, int m, int32_t *suff);
void oracle_pre_bm_gs(const uint8_t *P, int m, int32_t *gs);
void oracle_pre_kmp(const uint8_t *P, int m, int32_t *next /* m+1 */);
uint32_t oracle_pre_so(const uint8_t *P, int m, uint32_t S[ORACLE_SIGMA]);
void oracle_pre_bndm(const uint8_t *P, int m, uint32_t B[ORACLE_SIGMA]);

/* --- corpora --- */
/* glibc rand() TYPE_3 stream as consumed by src/textgen.c:34-54: one
 * continuous stream, 5,000,000 draws per corpus in the order
 * sigma = 2,4,8,16,32,64,128,250.  Writes the first `n` bytes (n <= 5e6) of
 * corpus rand<sigma>; returns 0, or -1 for a sigma textgen does not emit. */
int oracle_textgen(int sigma, uint8_t *out, uint64_t n);

/* Counter-based generator for the GPU-scale configs:
 * byte i = (splitmix64(seed + (i>>3)) >> (8*(i&7))) & 0xFF, reduced to
 * [0,sigma) by mask (power of two) or modulo.  Fills out[0..n) with text
 * bytes off..off+n. */
void oracle_gen_text(uint64_t seed, int sigma, uint64_t off, uint64_t n,
                     uint8_t *out);
uint64_t oracle_splitmix64(uint64_t x);

/* Multi-core driver for the cpu_baseline leg: splits [0, n-m] start
 * positions over `threads` pthreads with an (m-1)-byte overlap and sums. */
uint64_t oracle_search_mt(const char *name, const uint8_t *P, int m,
                          const uint8_t *T, uint64_t n, int threads);

#ifdef __cplusplus
}
#endif
#endif
