/*
 * test — correctness harness for one algorithm of the MI355X engine.
 *
 * Restates SMART's tester (reference: src/test.c): known-answer cases checked
 * against an in-file brute force (test.c:45-56), through the same entry point
 * the benchmark driver uses — here the C ABI's plugin-shaped
 * smartgpu_<algo>_search(P,m,T,n) instead of a spawned executable (test.c:67-74).
 *   ./test ALGONAME [-nv]      exit status 0 = passed, 1 = failed
 * Cases 1-20 follow test.c:252-382 (12-15 use a fixed seed instead of
 * srand(time)); cases 21+ close gaps of the reference suite:
 * n > 64, m > 40 up to XSIZE, sigma = 256, matches at both text ends, a
 * would-be match straddling the text end.  Unlike test.c:99, a failing (-1)
 * search is a failure.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "smartgpu.h"

typedef int (*search_fn)(const unsigned char *, int, const unsigned char *, int);

static int brute_force(const unsigned char *x, int m, const unsigned char *y, int n)
{
    int count = 0;
    for (int j = 0; j + m <= n; ++j) {
        int i = 0;
        while (i < m && x[i] == y[j + i]) ++i;
        count += (i == m);
    }
    return count;
}

static int verbose = 1, failures = 0, cases = 0;
static int min_m = 1; /* shortest pattern the algorithm applies to (raita.c:37: 2); below it -1 is the expected answer */
static search_fn algo;

static void attempt(int no, const unsigned char *P, int m, const unsigned char *T, int n)
{
    ++cases;
    int want = brute_force(P, m, T, n);
    int got = algo(P, m, T, n);
    if (m < min_m) want = -1;
    if (got != want) {
        ++failures;
        if (verbose) printf("\n\tERROR: test failed on case n.%d (m=%d, n=%d)\n\t\tfound %d occ instead of %d\n\n", no, m, n, got, want);
    }
}

static void attempt_str(int no, const char *p, const char *t)
{
    attempt(no, (const unsigned char *)p, (int)strlen(p), (const unsigned char *)t, (int)strlen(t));
}

static unsigned lcg_state = 12345u;
static unsigned lcg(void) { return (lcg_state = lcg_state * 1103515245u + 12345u) >> 16; }

int main(int argc, char **argv)
{
    if (argc == 1) {
        printf("\n\tSMART UTILITY FOR TESTING STRING MATCHING ALGORITHMS (MI355X engine)\n\n\tusage: ./test ALGONAME [-nv]\n\n");
        return 0;
    }
    if (argc > 2 && !strcmp(argv[2], "-nv")) verbose = 0;
    switch (smartgpu_algo_id(argv[1])) {
        case SMARTGPU_HOR: algo = smartgpu_hor_search; break;
        case SMARTGPU_BM: algo = smartgpu_bm_search; break;
        case SMARTGPU_KMP: algo = smartgpu_kmp_search; break;
        case SMARTGPU_SO: algo = smartgpu_so_search; break;
        case SMARTGPU_BNDM: algo = smartgpu_bndm_search; break;
        case SMARTGPU_EPSM: algo = smartgpu_epsm_search; break;
        case SMARTGPU_SA: algo = smartgpu_sa_search; break;
        case SMARTGPU_QS: algo = smartgpu_qs_search; break;
        case SMARTGPU_TUNEDBM: algo = smartgpu_tunedbm_search; break;
        case SMARTGPU_RAITA: algo = smartgpu_raita_search; min_m = 2; break;
        case SMARTGPU_HASH3: algo = smartgpu_hash3_search; min_m = 3; break;
        case SMARTGPU_HASH5: algo = smartgpu_hash5_search; min_m = 5; break;
        case SMARTGPU_HASH8: algo = smartgpu_hash8_search; min_m = 8; break;
        case SMARTGPU_SBNDM: algo = smartgpu_sbndm_search; min_m = 2; break;
        case SMARTGPU_KR: algo = smartgpu_kr_search; break;
        case SMARTGPU_BNDML: algo = smartgpu_bndml_search; break;
        default: printf("\tunknown algorithm %s\n", argv[1]); return 1;
    }
    if (smartgpu_device_count() < 1) { fprintf(stderr, "test: no GPU: %s\n", smartgpu_last_error()); return 1; }
    if (verbose) { printf("\n\tPlease, wait a moment.............."); fflush(stdout); }

    /* 1-11: fixed strings, n = 10 (test.c:252-316) */
    attempt_str(1, "a", "aaaaaaaaaa");
    attempt_str(2, "aa", "aaaaaaaaaa");
    attempt_str(3, "aaaaaaaaaa", "aaaaaaaaaa");
    attempt_str(4, "b", "aaaaaaaaaa");
    attempt_str(5, "ab", "ababababab");
    attempt_str(6, "a", "ababababab");
    attempt_str(7, "aba", "ababababab");
    attempt_str(8, "abc", "ababababab");
    attempt_str(9, "ba", "ababababab");
    attempt_str(10, "babbbbb", "ababababab");
    attempt_str(11, "bcdefg", "bcdefghilm");
    /* 12-15: random sigma=128 (test.c:318-344) */
    static unsigned char T[70000], P[SMARTGPU_XSIZE + 8];
    for (int c = 12; c <= 13; ++c) {
        for (int h = 0; h < 10; ++h) T[h] = (unsigned char)(lcg() % 128);
        memcpy(P, T, 4);
        attempt(c, P, 4, T, 10);
    }
    for (int c = 14; c <= 15; ++c) {
        for (int h = 0; h < 64; ++h) T[h] = (unsigned char)(lcg() % 128);
        memcpy(P, T, 40);
        attempt(c, P, 40, T, 64);
    }
    /* 16-18: unary and (ab)^k texts, m = 40 (test.c:346-370) */
    memset(T, 'a', 64);
    memset(P, 'a', 40);
    attempt(16, P, 40, T, 64);
    for (int h = 0; h < 64; ++h) T[h] = (h & 1) ? 'b' : 'a';
    for (int h = 0; h < 40; ++h) P[h] = (h & 1) ? 'b' : 'a';
    attempt(17, P, 40, T, 64);
    P[39] = 'c';
    attempt(18, P, 40, T, 64);
    /* 19-20 (test.c:372-382) */
    attempt_str(19, "babbbbb", "abababbbbb");
    attempt_str(20, "bababb", "abababbbbb");

    /* 21+: beyond the reference suite */
    int no = 21;
    const int sigmas[] = {2, 4, 128, 256};
    const int lens[] = {1, 2, 3, 4, 5, 8, 16, 31, 32, 33, 64, 65, 255, 256, 257, 1000, 4096, 4200};
    for (unsigned si = 0; si < sizeof sigmas / sizeof *sigmas; ++si) {
        const int n = 65536 + 13;
        for (int h = 0; h < n; ++h) T[h] = (unsigned char)(lcg() % (unsigned)sigmas[si]);
        for (unsigned li = 0; li < sizeof lens / sizeof *lens; ++li) {
            const int m = lens[li];
            memcpy(P, T + 777, (size_t)m);          /* somewhere inside */
            attempt(no++, P, m, T, n);
            memcpy(P, T, (size_t)m);                /* at the very start */
            attempt(no++, P, m, T, n);
            memcpy(P, T + n - m, (size_t)m);        /* at the very end */
            attempt(no++, P, m, T, n);
            P[m / 2] ^= 1;                          /* near miss */
            attempt(no++, P, m, T, n);
            /* a match that would need bytes past the end: search n-1 bytes for the end pattern */
            memcpy(P, T + n - m, (size_t)m);
            attempt(no++, P, m, T, n - 1);
        }
    }
    memset(T, 'a', 70000);                          /* dense overlaps, long text */
    memset(P, 'a', 4200);
    attempt(no++, P, 1, T, 70000);
    attempt(no++, P, 33, T, 70000);
    attempt(no++, P, 4200, T, 70000);
    attempt(no++, P, 5, T, 4);                      /* m > n */

    if (failures) {
        if (verbose) printf("\n\t%d of %d cases FAILED\n\n", failures, cases);
        return 1;
    }
    if (verbose) printf("\n\tWell done! Test passed successfully (%d cases)\n\n", cases);
    return 0;
}
