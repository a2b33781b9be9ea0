/*
 * smartgpu.h — C ABI of the MI355X exact-string-matching engine that sits
 * behind SMART's per-algorithm plugin surface.
 *
 * Drop-in boundary.  Each entry point names the reference
 * interface it replaces (paths relative to the SMART tree):
 *
 *   reference                                      this library
 *   ---------------------------------------------  ---------------------------------
 *   int search(unsigned char*,int,unsigned char*,  smartgpu_<algo>_search()   (same shape,
 *       int)          src/algos/include/main.h:39    same return convention: count, -1 = n/a)
 *   double *run_time,*pre_time (ms, written by the  smartgpu_last_times(), and the pre_ms /
 *       BEGIN_/END_ macros)   main.h:28-31,34-35    run_ms out-params of smartgpu_search64()
 *   text in a SysV segment: shmget(tkey,TSIZE+10)   smartgpu_text_upload()/_generate()/_free():
 *       + getText()    src/smart.c:553-568,95-138    the text lives in HBM for a whole run
 *   execute(): system("./source/bin/<algo> shared   smartgpu_search64() — an in-process call
 *       ...")          src/smart.c:140-146           instead of a process spawn per pattern
 *   textgen rand-sigma corpora  src/textgen.c:34-54 smartgpu_text_generate() (on-device,
 *                                                    counter-based)
 *
 * All pointers are plain host pointers unless a parameter says "device"; there
 * are no C++ or torch types in any signature.  The library is single-threaded
 * from the caller's point of view (like SMART); every call that returns a count
 * is synchronous.  Errors: negative return codes, text in smartgpu_last_error().
 * Threading contract: ONE host thread per device at a time — the per-device stream, staging buffer and table arena are
 * shared by every call on that device (src/smart.c is single-threaded); last_error / last_times are
 * thread-local.  Loading the library sets HSA_ENABLE_IPC_MODE_LEGACY=0 (unless the environment already has it) so that
 * RCCL works on hosts that only support dmabuf IPC: load it before the process makes its first HIP call.
 *
 * There is no CPU fallback: without a usable HIP device every compute entry
 * point fails with SMARTGPU_ERR_HIP.
 */
#ifndef SMARTGPU_H
#define SMARTGPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SMARTGPU_XSIZE 4200 /* longest pattern, src/algos/include/define.h:25 */

/* return codes (SMART keeps count>=0 / -1 "not applicable"; src/smart.c:143-145,330-343) */
#define SMARTGPU_OK 0
#define SMARTGPU_NA (-1)          /* algorithm not applicable / count does not fit an int */
#define SMARTGPU_ERR_ARG (-3)     /* bad argument (unknown algorithm, m<1, m>XSIZE, range) */
#define SMARTGPU_ERR_HIP (-4)     /* HIP runtime error or no device */
#define SMARTGPU_ERR_NOMEM (-5)

/* algorithm ids; names are SMART's lower-case executable names (src/smart.c:142) */
enum {
    SMARTGPU_HOR = 0,  /* src/algos/hor.c  */
    SMARTGPU_BM = 1,   /* src/algos/bm.c   */
    SMARTGPU_KMP = 2,  /* src/algos/kmp.c  */
    SMARTGPU_SO = 3,   /* src/algos/so.c   */
    SMARTGPU_BNDM = 4, /* src/algos/bndm.c */
    SMARTGPU_EPSM = 5, /* src/algos/epsm.c */
    /* adjacent algorithm names on the same engine */
    SMARTGPU_SA = 6,      /* src/algos/sa.c      Shift-And, the dual of so.c            */
    SMARTGPU_QS = 7,      /* src/algos/qs.c      Quick Search: shift on the byte after the window */
    SMARTGPU_TUNEDBM = 8, /* src/algos/tunedbm.c Horspool's table with a zero entry + skip loop   */
    SMARTGPU_RAITA = 9,   /* src/algos/raita.c   Horspool's shifts, last/middle/first/rest order; m >= 2 (raita.c:37) */
    SMARTGPU_HASH3 = 10,  /* src/algos/hash3.c   Lecroq: shift under an 8-bit hash of the last 3 bytes; m >= 3 */
    SMARTGPU_HASH5 = 11,  /* src/algos/hash5.c   ... of the last 5 bytes; m >= 5 */
    SMARTGPU_HASH8 = 12,  /* src/algos/hash8.c   ... of the last 8 bytes; m >= 8 */
    SMARTGPU_SBNDM = 13,  /* src/algos/sbndm.c   Simplified BNDM; m >= 2 */
    SMARTGPU_KR = 14,     /* src/algos/kr.c      Karp-Rabin: rolling 32-bit hash + confirmation */
    SMARTGPU_BNDML = 15,  /* src/algos/bndml.c   BNDM with multi-word bit vectors for m > 32 */
    SMARTGPU_NUM_ALGOS = 16
};

typedef struct smartgpu_text smartgpu_text; /* a text resident in one GPU's HBM */
typedef struct smartgpu_plan smartgpu_plan; /* one (algorithm, pattern) with its tables in HBM */

/* ---- library ---------------------------------------------------------- */
const char *smartgpu_version(void);
const char *smartgpu_last_error(void);
int smartgpu_device_count(void);                   /* <0 on error */
int smartgpu_algo_id(const char *name);            /* "hor","bm","kmp","so","bndm","epsm","sa","qs","tunedbm","raita","hash3","hash5","hash8","sbndm","kr","bndml" (any case); -1 unknown */
const char *smartgpu_algo_name(int algo);          /* NULL if out of range */
int smartgpu_device_sync(int device);              /* waits for the library's stream on `device` */

/* ---- text lifecycle (replaces shmget + getText, src/smart.c:553-568,95-138) ---- */
/* Copies host[0..n) into HBM of `device` through pinned staging.  NULL on error. */
smartgpu_text *smartgpu_text_upload(const void *host, uint64_t n, int device);
/* Text byte i = unit[(phase + i) % unit_len] for i in [0,n): a corpus replicated
 * to a target size (BASELINE config 4) without shipping n bytes over PCIe. */
smartgpu_text *smartgpu_text_upload_tiled(const void *unit, uint64_t unit_len, uint64_t phase,
                                          uint64_t n, int device);
/* Text byte i = byte (off+i) of the counter-based rand-sigma corpus, generated on
 * the device: splitmix64(seed + (j>>3)) >> (8*(j&7)), masked (sigma a power of
 * two) or reduced modulo sigma.  2 <= sigma <= 256. */
smartgpu_text *smartgpu_text_generate(uint64_t seed, int sigma, uint64_t off, uint64_t n, int device);
void smartgpu_text_free(smartgpu_text *t);
uint64_t smartgpu_text_length(const smartgpu_text *t);
int smartgpu_text_device(const smartgpu_text *t);
/* Copies text[off..off+len) back to the host (tests, pattern extraction à la
 * setOfRandomPatterns, src/smart.c:148-158). */
int smartgpu_text_read(const smartgpu_text *t, uint64_t off, uint64_t len, void *host);
/* Which byte values the text holds: bit c of bits[8] (bit c%32 of word c/32) is set iff some text byte equals c.
 * Taken once, on the device, when the text is created (the reference's getText, src/smart.c:95-138, reads the corpus
 * once per run as well; a text is never written afterwards).  The runs kernels use it: on a text of at most four
 * distinct values they take four bytes per table step. */
int smartgpu_text_alphabet(const smartgpu_text *t, uint32_t bits[8]);

/* ---- searching -------------------------------------------------------- */
/* Counts the occurrences of P[0..m) whose window lies inside text[off..off+n),
 * i.e. start positions s in [off, off+n-m].  Overlapping occurrences count
 * (define.h:33).  *pre_ms = host table construction + table upload
 * (BEGIN_/END_PREPROCESSING, main.h:28,30); *run_ms = kernel(s) + count readback,
 * by HIP events (BEGIN_/END_SEARCHING, main.h:29,31).  Either may be NULL. */
int smartgpu_search64(int algo, const uint8_t *P, uint32_t m, const smartgpu_text *text,
                      uint64_t off, uint64_t n, uint64_t *count, double *pre_ms, double *run_ms);

/* The harness's inner loop as ONE call (src/smart.c:312-345: for each of the -pset patterns of a length,
 * execute() and read count and times back): K patterns of m bytes each, P[0..K), over the same resident
 * text range.  Preprocessing: the K tables are built on the host and placed in one arena in HBM (grown
 * when a batch needs more, never allocated per pattern); searching: K launches back to back on the
 * device's stream and ONE read-back of the K counts, so the synchronous per-call cost of
 * smartgpu_search64 (25-30 us) is paid once per pattern set.  On texts up to 32 MiB the patterns whose plans
 * choose the same kernel share one grid (gridDim.y = pattern, at most 65535 per grid: a larger group is
 * launched in slices).  K is at most 262144 (2^18) per call: SMARTGPU_ERR_ARG beyond.
 *   counts[k]   occurrences of P[k]                                   (K entries, required)
 *   pre_ms[k]   host table construction of P[k] + its share of the upload     (K entries or NULL)
 *   run_ms[k]   device time of the k-th search by HIP events — one event per pattern — (K entries or NULL:
 *               no per-pattern events)
 *   *batch_ms   wall clock from the first launch to the counts on the host   (or NULL)              */
int smartgpu_search_batch64(int algo, const uint8_t *const *P, uint32_t m, uint32_t K, const smartgpu_text *text,
                            uint64_t off, uint64_t n, uint64_t *counts, double *pre_ms, double *run_ms,
                            double *batch_ms);
/* The same, with EVERY pattern launched on its own between its own pair of events, whatever the text's size: run_ms[k]
 * is then pattern k's device time alone (in the one-grid form above a group's patterns share their group's time).  What
 * the harness uses when best / worst / standard deviation or the -tb bound are asked for: src/smart.c:320-329 times
 * every pattern, :337-343 applies the bound per run, :347-351 derives best, worst and std from those times. */
int smartgpu_search_batch64_each(int algo, const uint8_t *const *P, uint32_t m, uint32_t K, const smartgpu_text *text,
                            uint64_t off, uint64_t n, uint64_t *counts, double *pre_ms, double *run_ms,
                            double *batch_ms);

/* SMART's own plugin shape, one symbol per algorithm (main.h:39).  T is a HOST
 * pointer: the text is uploaded for the call and released afterwards, so this
 * is the compatibility path, not the fast one.  Returns the count, or -1 when
 * it does not fit an int / on error (smart.c:143-145 maps any failure to -1). */
int smartgpu_hor_search(const unsigned char *P, int m, const unsigned char *T, int n);
int smartgpu_bm_search(const unsigned char *P, int m, const unsigned char *T, int n);
int smartgpu_kmp_search(const unsigned char *P, int m, const unsigned char *T, int n);
int smartgpu_so_search(const unsigned char *P, int m, const unsigned char *T, int n);
int smartgpu_bndm_search(const unsigned char *P, int m, const unsigned char *T, int n);
int smartgpu_epsm_search(const unsigned char *P, int m, const unsigned char *T, int n);
int smartgpu_sa_search(const unsigned char *P, int m, const unsigned char *T, int n);      /* sa.c:36-94 */
int smartgpu_qs_search(const unsigned char *P, int m, const unsigned char *T, int n);      /* qs.c:33-52 */
int smartgpu_tunedbm_search(const unsigned char *P, int m, const unsigned char *T, int n); /* tunedbm.c:36-65 */
int smartgpu_raita_search(const unsigned char *P, int m, const unsigned char *T, int n);   /* raita.c:35-64; -1 for m < 2 */
int smartgpu_hash3_search(const unsigned char *P, int m, const unsigned char *T, int n);   /* hash3.c:28-84; -1 for m < 3 */
int smartgpu_hash5_search(const unsigned char *P, int m, const unsigned char *T, int n);   /* hash5.c; -1 for m < 5 */
int smartgpu_hash8_search(const unsigned char *P, int m, const unsigned char *T, int n);   /* hash8.c; -1 for m < 8 */
int smartgpu_sbndm_search(const unsigned char *P, int m, const unsigned char *T, int n);   /* sbndm.c:28-149; -1 for m < 2 */
int smartgpu_kr_search(const unsigned char *P, int m, const unsigned char *T, int n);      /* kr.c:28-54 */
int smartgpu_bndml_search(const unsigned char *P, int m, const unsigned char *T, int n);   /* bndml.c:44-132 */
/* pre/run times (ms) of the last search on this thread (main.h:34-35 globals) */
void smartgpu_last_times(double *pre_ms, double *run_ms);

/* ---- occurrence positions (extension) -------------------------- */
/* The reference only counts (OUTPUT(j) is count++, define.h:33).  This call also returns WHERE:
 * every s in [off, off+n-m] with T[s..s+m) == P, ascending, relative to text byte 0, through the
 * packed matcher with an output stage.  positions is a HOST buffer of `cap` entries.
 * *count always receives the number of occurrences.  Returns SMARTGPU_OK when count <= cap (the
 * list is complete), SMARTGPU_ERR_NOMEM when it is not (retry with cap >= *count). */
int smartgpu_find64(const uint8_t *P, uint32_t m, const smartgpu_text *text, uint64_t off, uint64_t n,
                    uint64_t *positions, uint64_t cap, uint64_t *count);

/* ---- plans: preprocess once, launch many (harness hot loop, smart.c:312-345) ---- */
/* Builds the algorithm's tables on the host and places them in HBM of `device`. */
smartgpu_plan *smartgpu_plan_create(int algo, const uint8_t *P, uint32_t m, int device);
void smartgpu_plan_free(smartgpu_plan *p);
/* Enqueues one search of text[off..off+n) on the device's stream and returns
 * without waiting; the count is ADDED to result slot `slot` (0 <= slot < 4096)
 * of the plan (slots start at zero; smartgpu_plan_reset() zeroes them again).
 * With `timed` != 0 the launch is bracketed by HIP events. */
int smartgpu_plan_launch(smartgpu_plan *p, const smartgpu_text *text, uint64_t off, uint64_t n,
                         int slot, int timed);
/* Waits for the stream and returns the count of `slot` (and, if the launch was
 * timed, its device time in ms; else *kernel_ms = -1). */
int smartgpu_plan_result(smartgpu_plan *p, int slot, uint64_t *count, double *kernel_ms);
/* Name of the dominant kernel the plan launches (as rocprofv3 reports it). */
const char *smartgpu_plan_kernel_name(const smartgpu_plan *p);
/* The same for (algo, P, m) without a device: which kernel a plan of this pattern would launch under the
 * current smartgpu_tune() settings — the host-side choice (DESIGN.md §4 "The plan reads the pattern").  NULL
 * if algo or m is out of range. */
const char *smartgpu_kernel_for(int algo, const uint8_t *P, uint32_t m);
/* Device address of the plan's uint64 result slots (for an RCCL reduce issued
 * by the caller on the same device). */
void *smartgpu_plan_result_device_ptr(smartgpu_plan *p);
/* Zeroes every result slot (stream-ordered). */
int smartgpu_plan_reset(smartgpu_plan *p);
/* Makes the plan write its counts to caller-owned DEVICE memory (`nslots`
 * uint64, zeroed by the caller), e.g. one element of a vector that the caller
 * reduces across GPUs with RCCL.  NULL restores the plan's own slots. */
int smartgpu_plan_set_result_buffer(smartgpu_plan *p, void *device_u64, int nslots);

/* ---- one process, several GPUs (the 8 GPUs of a node) ---------------------------- */
/* A text sharded by byte offset over `ngpus` devices of this process: device g owns the
 * start positions [g*n/k, (g+1)*n/k) and holds SMARTGPU_XSIZE extra bytes, so any pattern
 * length can be searched without an exchange step.  `devices` lists the
 * device ordinals (NULL = 0..ngpus-1). */
typedef struct smartgpu_mtext smartgpu_mtext;
smartgpu_mtext *smartgpu_mtext_upload(const void *host, uint64_t n, int ngpus, const int *devices);
smartgpu_mtext *smartgpu_mtext_generate(uint64_t seed, int sigma, uint64_t n, int ngpus, const int *devices);
void smartgpu_mtext_free(smartgpu_mtext *t);
uint64_t smartgpu_mtext_length(const smartgpu_mtext *t);
int smartgpu_mtext_ngpus(const smartgpu_mtext *t);
/* The partition itself (pure arithmetic, no device): shard g of `ngpus` over a text of n bytes owns the start positions
 * [*begin, *begin + *own) — the shards' sizes differ by at most one byte and add up to n — and holds *held bytes from
 * *begin on: its own and up to SMARTGPU_XSIZE - 1 of the following shards', never beyond byte n.  Any pointer may be NULL. */
int smartgpu_mtext_partition(uint64_t n, int ngpus, int g, uint64_t *begin, uint64_t *own, uint64_t *held);
/* Self-test of the host-thread pool that enqueues the k devices' launches of a multi-GPU search at once (no device
 * needed): `rounds` rounds of up to k jobs, each job must run exactly once per round.  0 = passed. */
int smartgpu_selftest_launch_pool(int k, int rounds);
/* Searches every shard concurrently (one stream per device) and sums the shard counts.
 * reduce = SMARTGPU_REDUCE_RCCL: ncclAllReduce(sum, uint64) over the devices' streams (RCCL
 * over xGMI; the devices must be distinct), then one 8-byte read-back;
 * reduce = SMARTGPU_REDUCE_HOST: eight-byte read-backs added on the host (also allows the
 * same device to appear more than once, which is how the shard arithmetic is tested on a
 * one-GPU box).  *run_ms covers launches + reduction + read-back. */
#define SMARTGPU_REDUCE_RCCL 0
#define SMARTGPU_REDUCE_HOST 1
int smartgpu_msearch64(int algo, const uint8_t *P, uint32_t m, smartgpu_mtext *text, int reduce,
                       uint64_t *count, double *pre_ms, double *run_ms);

/* The pattern-set form of smartgpu_msearch64: every device searches its shard for all K patterns, then ONE
 * reduction of the K counts (RCCL: one ncclAllReduce of K uint64 per device, in place) and one read-back. */
int smartgpu_msearch_batch64(int algo, const uint8_t *const *P, uint32_t m, uint32_t K, smartgpu_mtext *text,
                             int reduce, uint64_t *counts, double *pre_ms, double *batch_ms);

/* ---- stream timing (hipEvents on the stream the kernels run on) ------------ */
int smartgpu_stream_mark(int device, int which /* 0 = begin, 1 = end */);
int smartgpu_stream_elapsed_ms(int device, double *ms); /* waits for mark 1 */
void *smartgpu_stream_handle(int device);               /* hipStream_t of the library on `device` */

/* Measures the device's practical streaming-read rate on this text (a plain
 * coalesced read-and-fold kernel, `reps` passes): the "measured streaming read"
 * the scan kernels are compared with besides the 8 TB/s spec peak. */
int smartgpu_probe_read_ms(const smartgpu_text *t, int reps, double *ms_per_pass);

/* Kernel-variant selection for experiments and A/B measurements (not needed in normal use;
 * every variant is parity-tested).  Keys:
 *   0  regime of the skip algorithms: 0 auto — short patterns and patterns whose symbols repeat on
 *      the packed matcher, patterns of 16+ bytes over two or three symbols on so_runs, the rest on
 *      the algorithm's own LDS-tile skip loop / 1 always the algorithm's own skip
 *      loop / 2 Horspool's bank-private LDS layout / 3 always the packed matcher
 *   1  bndm_scan: bytes of a window read per iteration (1, 2, 4, 8; 0 = the plan's choice from the pattern);
 *      9 = the plan's choice and never the gram form of texts of at most four byte values (round 4)
 *   2  bm_scan / bndm_scan workgroups: 1 four waves / 2 two waves (0 = default: bm_scan two where the pattern's symbols
 *      repeat, bndm_scan always four); 3 = Horspool's nested loop also where the pattern's symbols repeat (default
 *      there: its flat form, round 3); 4 = Horspool, Tuned BM and Boyer-Moore never on grams (round 4: hor_scan_gram,
 *      bm_scan_gram on texts of at most four byte values)
 *   4  workgroups per CU of the LDS-tile kernels (0 = the launcher's choice)
 *   3  KMP: 0 kmp_runs (transition table) / 1 kmp_scan (LDS tiles, m <= 40) / 2 kmp_links_runs
 *      (failure links followed per byte) / 5 kmp_runs a byte per table step even on a text of at most four
 *      byte values (round 3: there it takes four) / 6 round 3's one-workgroup-per-CU form (round 4: five compact
 *      workgroups per CU; key 4 sets their number)
 *   5  run length in bytes of the runs kernels (so_runs, kmp_runs); 0 = default
 *   6  SO: 0 so_runs (bank-private table, line fetch) / 1 so_scan (LDS tiles) / 2 so_runs64
 *      (shared table, 64-byte steps); SA: 3 = its own AND form (default: the complemented, Shift-Or form);
 *      5 so_runs a byte per table lookup even on a text of at most four byte values (round 3: there it takes four)
 *   7  packed matcher: 0 default — v_mqsad_pk_u16_u8 references for m <= 7 and on texts of at most four byte values,
 *      dword compares otherwise, the neighbour lane's bytes by DPP (round 4) / 1 both loads cached / 3 one load +
 *      shuffle / 6 v_mqsad references at every length / 7 the second (cached) load instead of DPP / 8, 9 dword
 *      compares only
 * Settings whose kernels exist only in the A/B build (libsmartgpu_ab.so) are refused by the product library. */
int smartgpu_tune(int key, int value);

/* Host-side preprocessing exposed for tests (same tables the kernels stage in
 * LDS): writes up to `cap` 32-bit entries, returns the number written or <0.
 *   which: 0 Horspool bad-char (256)      hor.c:26-30 / bm.c:27-33
 *          1 BM good-suffix (m)           bm.c:36-66
 *          2 KMP failure function (m+1)   kmp.c:27-41
 *          3 Shift-Or masks (256)         so.c:27-38   (32-bit words, prefix of 32 for m>32)
 *          4 BNDM masks (256)             bndm.c:35-40 (same)
 *          5 KMP transition table ((m+1)*256, m <= 255): the failure links of kmp.c:27-41
 *            expanded into delta[state][byte]; state m = an occurrence ends here
 *          6 the same over the pattern's own alphabet: k1, colmap[256], table[(m+1)*k1]
 *          7 Shift-And masks S[256] (sa.c:27-34), 8 Quick Search shifts qsBc[256] (qs.c:27-31),
 *          9 kmp_runs' tables as the kernel holds them in LDS (bytes; the last 272: Q and thr),
 *          10 the two-bit codes of the byte values of P taken as a SET (what the runs kernels use on a text of at most
 *            four byte values): shift, symtab — code (c >> shift) & 3, byte `code` of symtab = the member with that code
 *            (a non-member with that code if none); 0 entries when there are more than four or no shift separates them,
 *          13/15/18 HASH3/5/8 shifts[256] followed by the shift after a candidate (hash3.c:36-56)   */
int smartgpu_build_table(int which, const uint8_t *P, uint32_t m, int32_t *out, uint32_t cap);

#ifdef __cplusplus
}
#endif
#endif /* SMARTGPU_H */
