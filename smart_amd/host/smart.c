/*
 * smart — benchmark driver for the MI355X exact-matching engine.
 *
 * Restates the behaviour of SMART's driver (reference: src/smart.c) on top of
 * the C ABI in include/smartgpu.h: same flags, same experiment loop, same
 * stdout report and TXT table.  What changes is the plumbing underneath:
 *   - the text lives in HBM (smartgpu_text_*) instead of a SysV segment
 *     (smart.c:553-568) and is loaded once per corpus (getText, smart.c:95-138);
 *   - an algorithm is an in-process call (smartgpu_search64) instead of
 *     system("./source/bin/<algo> shared ...") (smart.c:140-146);
 *   - pre/search times come back as out-parameters instead of two 8-byte shm
 *     segments (main.h:28-37).
 * Report files: TXT (-txt), LaTeX (-tex), XML (always) and the PHP data array (-php) in the
 * formats of src/output.h:49-247; the HTML page of every corpus (always) with its charts drawn
 * as inline SVG — the reference's pages load the RGraph scripts of its results/ directory,
 * which are not part of this project — and results/<code>/index.html linking the pages
 * (outputINDEX, output.h:706-741).
 * Additions: -algo LIST, -data DIR, -gpu D, -gpus K, device-generated rand corpora when the
 * data directory has none; every report carries GB/s of text scanned, its share of the HBM-read
 * roofline (8 TB/s per GPU x the GPUs the text is sharded over) and the number of GPUs: columns of
 * the stdout line, <corpus>.roofline.txt beside the TXT table, <GPUS>/<GBS>/<ROOFLINE> in the XML.
 *
 * Build: make -C smart_amd/host   (gcc, links ../csrc/libsmartgpu.so)
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>

#include "smartgpu.h"

#define XSIZE SMARTGPU_XSIZE
#define MAX_ALGOS SMARTGPU_NUM_ALGOS
#define MAX_LENGTHS 17
#define MAX_CORPORA 15
#define MAX_RUNS 5000 /* -pset upper bound (smart.c:183 has STDTIME[5000]) */
#define HBM_PEAK_GBS 8000.0 /* MI355X HBM3E read peak per GPU: the roofline of a byte scan */

/* pattern-length sets (reference: src/sets.h:23-25) */
static const int LEN_VERY_SHORT[] = {1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 0};
static const int LEN_SHORT[] = {2, 4, 6, 8, 10, 12, 14, 16, 18, 20, 22, 24, 26, 28, 30, 32, 0};
static const int LEN_LARGE[] = {2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 0};

/* corpus names and alphabet sizes (reference: src/sets.h:26-27) */
static const char *CORPUS[MAX_CORPORA] = {"rand2", "rand4", "rand8", "rand16", "rand32", "rand64", "rand128",
                                          "rand250", "italianTexts", "englishTexts", "frenchTexts",
                                          "chineseTexts", "midimusic", "genome", "protein"};
static const int CORPUS_SIGMA[MAX_CORPORA] = {2, 4, 8, 16, 32, 64, 128, 250, 128, 128, 128, 128, 128, 64, 64};

#define BATCH 512 /* patterns per smartgpu_search_batch64 call (the default -pset in one): the -tb limit and an ERROR are looked at between calls */

/* per-search status, as the reference harness records it (smart.c:143-145,330-343) */
enum { ST_OK = 1, ST_ERROR = 0, ST_NA = -1, ST_OUT = -2 };

struct options {
    int runs;          /* -pset  (smart.c:415 VOLTE = 500) */
    long tsize;        /* -tsize in bytes (smart.c:416: 1 MiB) */
    int minlen, maxlen;
    const int *lengths;
    int occ, pre, dif, std, txt, tex, php;
    double limit_ms;   /* -tb (smart.c:424: 300 ms); fractions of a millisecond are accepted here (a search takes microseconds) */
    int each;          /* -dif, -std or -tb given: every pattern is launched and timed on its own (smart.c:320-351) */
    int device;
    int gpus;          /* -gpus k: shard the text over k GPUs of this process, RCCL sum of the counts */
    long seed;         /* -seed S: srandom(S) instead of the reference's srand(time(NULL)) (smart.c:432): repeatable pattern sets */
    int reduce;        /* SMARTGPU_REDUCE_RCCL, or _HOST when the environment says SMARTGPU_REDUCE_HOST=1 (rehearsal) */
    const char *data_dir;
    char text_arg[256];
    char simple_p[128], simple_t[1100];
    int simple;
    int algos[MAX_ALGOS], nalgos;
};

static void usage(void)
{
    /* the flags of the reference driver (src/smart.c:48-71), described in this project's words */
    printf("\tsmart - exact string matching benchmark on the MI355X engine\n\n");
    printf("\t-pset N       number of patterns per length; times are means over them (default 500)\n");
    printf("\t-tsize S      use at most S MiB of the corpus as the text (default 1)\n");
    printf("\t-plen L U     only pattern lengths in [L, U]\n");
    printf("\t-text F       corpus to search (required unless -simple): a name, \"all\", or a list A-B-C\n");
    printf("\t-short        pattern lengths 2,4,...,32 instead of 2,4,8,...,4096\n");
    printf("\t-vshort       pattern lengths 1..16\n");
    printf("\t-occ          also print the mean number of occurrences\n");
    printf("\t-pre          report preprocessing and searching times separately\n");
    printf("\t-tb L         give up on an algorithm when one search exceeds L ms (default 300; 0.25 is a valid L)\n");
    printf("\t-dif          also print the best and the worst time\n");
    printf("\t-std          also print the standard deviation\n");
    printf("\t              (a pattern set is searched in calls of up to %d patterns.  With -dif, -std or -tb every\n", BATCH);
    printf("\t              pattern is launched and timed on its own, as the reference times every pattern; without\n");
    printf("\t              them the patterns of a call that share a kernel share ONE grid on texts up to 32 MiB —\n");
    printf("\t              faster, and only the mean is reported then)\n");
    printf("\t-txt          write the result table as results/<code>/<corpus>.txt\n");
    printf("\t-tex          write it as a LaTeX tabular too\n");
    printf("\t-php          write it as a PHP array (results/<code>/<corpus>.php) too\n");
    printf("\t-simple P T   one search of pattern P (<= 100 chars) in text T (<= 1000 chars)\n");
    printf("\t-algo LIST    comma separated algorithms out of hor,bm,kmp,so,bndm,epsm,sa,qs,tunedbm,raita,\n");
    printf("\t              hash3,hash5,hash8,sbndm,kr,bndml (default: all of them, or the ones marked #1 in\n");
    printf("\t              source/algorithms.h when that file exists)\n");
    printf("\t-data DIR     directory holding <corpus>/index.txt (default \"data\")\n");
    printf("\t-gpu D        device ordinal (default 0)\n");
    printf("\t-gpus K       shard the text by byte offset over GPUs 0..K-1 (RCCL sum of the counts);\n");
    printf("\t              with SMARTGPU_REDUCE_HOST=1 in the environment the counts are added on the host and\n");
    printf("\t              K may exceed the visible GPUs (shard g on GPU g mod visible): a rehearsal of the\n");
    printf("\t              sharded path on a smaller box, its times mean nothing\n");
    printf("\t-seed S       draw the patterns from srandom(S) (default: the time, as the reference does)\n");
    printf("\t-h            this help\n\n");
}

/* The pattern offsets come from a generator state of this file's own (glibc random_r, the TYPE_3 generator random()
 * uses): libc's shared random()/rand() state is also drawn from by the GPU runtime's threads, so a seeded run
 * (-seed) was not repeatable through srandom()/random() (smart.c:432,153 use srand(time)/random()). */
static struct random_data draw_state;
static char draw_buf[128];
static void seed_draw(unsigned seed)
{
    memset(&draw_state, 0, sizeof draw_state);
    initstate_r(seed, draw_buf, sizeof draw_buf, &draw_state);
}
static long draw(void)
{
    int32_t v = 0;
    random_r(&draw_state, &v);
    return (long)v;
}

static int is_number(const char *s)
{
    if (!*s) return 0;
    for (; *s; ++s)
        if (*s < '0' || *s > '9') return 0;
    return 1;
}

static int is_decimal(const char *s) /* digits with at most one '.' */
{
    int digits = 0, dots = 0;
    for (; *s; ++s) {
        if (*s == '.') ++dots;
        else if (*s >= '0' && *s <= '9') ++digits;
        else return 0;
    }
    return digits > 0 && dots <= 1;
}

static void upper(char *dst, const char *src)
{
    for (; *src; ++src, ++dst) *dst = (*src >= 'a' && *src <= 'z') ? (char)(*src - 'a' + 'A') : *src;
    *dst = 0;
}

/* registry in SMART's format: one "#<0|1> #<name> " entry per algorithm
 * (reference: getAlgo, src/function.h:62-77).  Returns the number selected,
 * -1 when the file is absent. */
static int read_registry(const char *path, int *sel)
{
    FILE *fp = fopen(path, "r");
    if (!fp) return -1;
    int c, n = 0;
    while ((c = getc(fp)) != EOF) {
        if (c != '#') continue;
        int flag = getc(fp) - '0';
        getc(fp);
        getc(fp); /* " #" */
        char name[32];
        int j = 0;
        while ((c = getc(fp)) != EOF && c != ' ' && c != '\n' && j < 31) name[j++] = (char)c;
        name[j] = 0;
        int id = smartgpu_algo_id(name);
        if (flag == 1 && id >= 0) sel[n++] = id;
    }
    fclose(fp);
    return n;
}

static int sigma_of(const char *corpus)
{
    for (int i = 0; i < MAX_CORPORA; ++i)
        if (!strcmp(CORPUS[i], corpus)) return CORPUS_SIGMA[i];
    return 0;
}

/* Load up to `cap` bytes of the files listed as #name# in <dir>/<corpus>/index.txt
 * (reference: getText, src/smart.c:95-138).  Returns the byte count, -1 if there
 * is no index file. */
static long load_corpus(const char *dir, const char *corpus, unsigned char *T, long cap)
{
    char path[600];
    snprintf(path, sizeof path, "%s/%s/index.txt", dir, corpus);
    FILE *index = fopen(path, "r");
    if (!index) return -1;
    long n = 0;
    int c;
    while (n < cap && (c = getc(index)) != EOF) {
        if (c != '#') continue;
        char name[300];
        int j = 0;
        while ((c = getc(index)) != EOF && c != '#' && j < 299) name[j++] = (char)c;
        name[j] = 0;
        snprintf(path, sizeof path, "%s/%s/%s", dir, corpus, name);
        printf("\tLoading the file %s\n", path);
        FILE *in = fopen(path, "r");
        if (!in) {
            printf("\tError in loading text file %s\n", path);
            continue;
        }
        n += (long)fread(T + n, 1, (size_t)(cap - n), in);
        fclose(in);
    }
    fclose(index);
    return n;
}

static void alphabet_report(const unsigned char *T, long n)
{
    int seen[256] = {0}, nalpha = 0, maxcode = 0;
    for (long i = 0; i < n; ++i) {
        if (!seen[T[i]]++) nalpha++;
        if (T[i] > maxcode) maxcode = T[i];
    }
    printf("\tAlphabet of %d characters.\n", nalpha);
    printf("\tGreater chararacter has code %d.\n", maxcode);
}

static void top_edge(void)
{
    printf("\t");
    for (int i = 0; i < 60; ++i) putchar('_');
    printf("\n");
}

struct cell { double mean, pre, best, worst, std, gbs, roof; int status; char kernel[24]; }; /* roof: gbs / (GPUs x 8 TB/s) */

/* One corpus: every pattern length x every algorithm x `runs` patterns
 * (reference: run_setting, src/smart.c:178-402). */
static void run_corpus(const struct options *o, const char *corpus, const unsigned char *T, long n,
                       smartgpu_text *text, smartgpu_mtext *mtext, const char *code,
                       struct cell table[MAX_ALGOS][MAX_LENGTHS])
{
    unsigned char **pats = malloc(sizeof(*pats) * (size_t)o->runs);
    for (int i = 0; i < o->runs; ++i) pats[i] = malloc(XSIZE + 1);
    double *sample = malloc(sizeof(double) * (size_t)(o->runs + 1));
    uint64_t bcount[BATCH];
    double bpre[BATCH], brun[BATCH];

    for (int il = 0; o->lengths[il] > 0; ++il) {
        const int m = o->lengths[il];
        if (m < o->minlen || m > o->maxlen) continue;
        if (!o->simple && m >= n) continue;
        /* patterns are cut from the text at random offsets (smart.c:148-158) */
        for (int i = 0; i < o->runs; ++i) {
            if (o->simple) {
                memcpy(pats[i], o->simple_p, (size_t)m);
            } else {
                long k = draw() % (n - m);
                memcpy(pats[i], T + k, (size_t)m);
            }
            pats[i][m] = 0;
        }
        printf("\n");
        top_edge();
        if (!o->simple) printf("\tExperimental results on %s: %s\n", corpus, code);
        else printf("\tExperimental results on %s\n", (const char *)T);
        printf("\tSearching for a set of %d patterns with length %d\n", o->runs, m);
        printf("\tTesting %d algorithms\n\n", o->nalgos);

        for (int ia = 0; ia < o->nalgos; ++ia) {
            const int algo = o->algos[ia];
            char name[32], head[64];
            upper(name, smartgpu_algo_name(algo));
            snprintf(head, sizeof head, "\t - [%d/%d] %s ", ia + 1, o->nalgos, name);
            printf("%s", head);
            for (size_t i = strlen(head); i < 35; ++i) putchar('.');
            fflush(stdout);

            struct cell *c = &table[ia][il];
            memset(c, 0, sizeof *c);
            c->best = 999.0;
            long long total_occ = 0;
            int status = ST_OK;
            {   /* which kernel the plans of this pattern set launch (the first pattern speaks for the set) */
                const char *kn = smartgpu_kernel_for(algo, pats[0], (uint32_t)m);
                snprintf(c->kernel, sizeof c->kernel, "%s", kn ? kn : "-");
            }
            /* The reference's loop (smart.c:312-345) spawns one process per pattern; here the whole set goes
             * to the engine in calls of BATCH patterns (smartgpu_search_batch64: tables of the set in one
             * arena, launches back to back, one read-back).  A pattern's time e is its share — by device time,
             * HIP events — of the wall time of its call (+ its preprocessing unless -pre).  With -dif, -std or -tb
             * (o->each) every pattern has its own launch and event pair, so best, worst, std and the -tb bound mean
             * what smart.c:337-351 means; without them patterns that share a kernel share one grid on texts up to
             * 32 MiB and a group's time is divided evenly among its patterns (only the mean is printed then). */
            int done = 0;
            if (!mtext) {
                /* an untimed pass first, the whole set over (at most) the first MiB: code object loads of every kernel
                 * the set's plans launch, the arena, first-touch copies and event allocation are not the algorithm's
                 * time (the reference's per-pattern processes have no such state) */
                const int kw = o->runs < BATCH ? o->runs : BATCH;
                const uint64_t nw = (uint64_t)n < (1ull << 20) ? (uint64_t)n : (1ull << 20);
                (void)smartgpu_search_batch64(algo, (const uint8_t *const *)pats, (uint32_t)m, (uint32_t)kw, text, 0, nw, bcount, bpre, brun, NULL);
            }
            while (done < o->runs && status == ST_OK) {
                const int kb = o->runs - done < BATCH ? o->runs - done : BATCH;
                int perc = (100 * (done + kb)) / o->runs;
                printf(perc < 10 ? "\b\b\b\b[%d%%]" : perc < 100 ? "\b\b\b\b\b[%d%%]" : "\b\b\b\b[%d%%]", perc);
                fflush(stdout);
                double batch_ms = 0;
                int rc = mtext ? smartgpu_msearch_batch64(algo, (const uint8_t *const *)(pats + done), (uint32_t)m, (uint32_t)kb, mtext,
                                                          o->reduce, bcount, bpre, &batch_ms)
                               : (o->each ? smartgpu_search_batch64_each : smartgpu_search_batch64)(
                                     algo, (const uint8_t *const *)(pats + done), (uint32_t)m, (uint32_t)kb, text, 0, (uint64_t)n, bcount, bpre, brun, &batch_ms);
                double dev_sum = 0;
                if (rc == SMARTGPU_OK && !mtext)
                    for (int k = 0; k < kb; ++k) dev_sum += brun[k];
                for (int k = 0; k < kb && status == ST_OK; ++k) {
                    long long occur = rc == SMARTGPU_OK ? (long long)bcount[k] : -1;
                    /* the call's wall time, shared out by device time (equal shares over several GPUs) */
                    double run_ms = rc != SMARTGPU_OK ? 0 : (!mtext && dev_sum > 0) ? batch_ms * brun[k] / dev_sum : batch_ms / kb;
                    double pre_ms = rc == SMARTGPU_OK ? bpre[k] : 0;
                    double e = o->pre ? run_ms : run_ms + pre_ms; /* smart.c:323 */
                    sample[done + k + 1] = e;
                    c->mean += e;
                    c->pre += pre_ms;
                    if (e < c->best) c->best = e;
                    if (e > c->worst) c->worst = e;
                    total_occ += occur;
                    if (occur <= 0 && !o->simple) status = occur == 0 ? ST_ERROR : ST_NA; /* smart.c:330-336 */
                    else if (e > o->limit_ms) status = ST_OUT;                             /* smart.c:337-343 */
                }
                done += kb;
            }
            if (status != ST_OK) {
                c->mean = c->pre = 0;
            } else {
                c->mean /= o->runs;
                c->pre /= o->runs;
                for (int k = 1; k <= o->runs; ++k) c->std += (sample[k] - c->mean) * (sample[k] - c->mean);
                c->std = sqrt(c->std / o->runs); /* population std, smart.c:349-351 */
                c->gbs = c->mean > 0 ? (double)n / (c->mean * 1e-3) / 1e9 : 0;
                c->roof = c->gbs / (HBM_PEAK_GBS * (o->gpus > 1 ? o->gpus : 1));
            }
            c->status = status;
            if (status == ST_OK) {
                char data[64];
                printf("\b\b\b\b\b\b\b.[OK]  ");
                /* the reference prints %.2f ms; a search of a 1 MiB text takes microseconds here */
                if (o->pre) snprintf(data, sizeof data, c->mean < 1.0 ? "\t%.4f + %.4f ms" : "\t%.2f + %.2f ms", c->pre, c->mean);
                else snprintf(data, sizeof data, c->mean < 1.0 ? "\t%.4f ms" : "\t%.2f ms", c->mean);
                printf("%s", data);
                for (size_t i = strlen(data); i < 20; ++i) putchar(' ');
                if (o->dif) {
                    snprintf(data, sizeof data, c->mean < 1.0 ? " [%.4f, %.4f]" : " [%.2f, %.2f]", c->best, c->worst);
                    printf("%s", data);
                    for (size_t i = strlen(data); i < 20; ++i) putchar(' ');
                }
                if (o->std) {
                    snprintf(data, sizeof data, c->mean < 1.0 ? " std %.4f" : " std %.2f", c->std);
                    printf("%s", data);
                    for (size_t i = strlen(data); i < 15; ++i) putchar(' ');
                }
                if (o->occ) printf("\tocc %lld", total_occ / o->runs);
                /* GB/s of text scanned, its share of the HBM-read roofline of the GPUs used, #GPUs, kernel */
                printf("\t%.1f GB/s\t%.1f%% of %d x 8 TB/s\t%s", c->gbs, 100.0 * c->roof, o->gpus > 1 ? o->gpus : 1, c->kernel);
                printf("\n");
            } else if (status == ST_ERROR) {
                printf("\b\b\b\b\b\b\b\b.[ERROR] \n");
            } else if (status == ST_NA) {
                printf("\b\b\b\b\b.[--]  \n");
                fprintf(stderr, "%s m=%d: %s\n", name, m, smartgpu_last_error());
            } else {
                printf("\b\b\b\b\b\b.[OUT]  \n");
            }
        }
    }
    printf("\n");
    top_edge();
    for (int i = 0; i < o->runs; ++i) free(pats[i]);
    free(pats);
    free(sample);
}

/* results/<code>/<corpus>.txt: one row per algorithm, one column per length
 * (reference: outputTXT, src/output.h:116-151) */
static void write_txt(const struct options *o, const char *corpus, const char *code,
                      struct cell table[MAX_ALGOS][MAX_LENGTHS])
{
    char path[400];
    mkdir("results", 0775);
    snprintf(path, sizeof path, "results/%s", code);
    mkdir(path, 0775);
    snprintf(path, sizeof path, "results/%s/%s.txt", code, corpus);
    FILE *fp = fopen(path, "w");
    if (!fp) return;
    for (int ia = 0; ia < o->nalgos; ++ia) {
        char name[32];
        upper(name, smartgpu_algo_name(o->algos[ia]));
        fprintf(fp, "%-20s", name);
        for (int il = 0; o->lengths[il] > 0; ++il) {
            int m = o->lengths[il];
            if (m < o->minlen || m > o->maxlen) continue;
            if (table[ia][il].mean > 0) fprintf(fp, "\t%.2f", table[ia][il].mean);
            else fprintf(fp, "\t-");
        }
        fprintf(fp, "\n");
    }
    fclose(fp);
    printf("\tOUTPUT RUNNING TIMES %s (results/%s/%s.txt)\n", code, code, corpus);
    /* the same table with the kernel each cell's plans launched (the reference's TXT format has no room for it) */
    snprintf(path, sizeof path, "results/%s/%s.kernels.txt", code, corpus);
    fp = fopen(path, "w");
    if (!fp) return;
    for (int ia = 0; ia < o->nalgos; ++ia) {
        char name[32];
        upper(name, smartgpu_algo_name(o->algos[ia]));
        fprintf(fp, "%-20s", name);
        for (int il = 0; o->lengths[il] > 0; ++il) {
            int m = o->lengths[il];
            if (m < o->minlen || m > o->maxlen) continue;
            fprintf(fp, "\t%s", table[ia][il].mean > 0 ? table[ia][il].kernel : "-");
        }
        fprintf(fp, "\n");
    }
    fclose(fp);
    /* and with each cell's share of the HBM-read roofline: a header line (GPUs, peak), then the table's shape */
    snprintf(path, sizeof path, "results/%s/%s.roofline.txt", code, corpus);
    fp = fopen(path, "w");
    if (!fp) return;
    fprintf(fp, "# GPUs %d\tHBM peak %.0f GB/s per GPU\tcells: %% of GPUs x peak (GB/s of text scanned)\n", o->gpus > 1 ? o->gpus : 1, HBM_PEAK_GBS);
    for (int ia = 0; ia < o->nalgos; ++ia) {
        char name[32];
        upper(name, smartgpu_algo_name(o->algos[ia]));
        fprintf(fp, "%-20s", name);
        for (int il = 0; o->lengths[il] > 0; ++il) {
            int m = o->lengths[il];
            if (m < o->minlen || m > o->maxlen) continue;
            if (table[ia][il].mean > 0) fprintf(fp, "\t%.1f%% (%.1f)", 100.0 * table[ia][il].roof, table[ia][il].gbs);
            else fprintf(fp, "\t-");
        }
        fprintf(fp, "\n");
    }
    fclose(fp);
}

/* results/<code>/<corpus>.tex: the same table as a LaTeX tabular
 * (reference: outputLatex, src/output.h:153-194) */
static void write_tex(const struct options *o, const char *corpus, const char *code,
                      struct cell table[MAX_ALGOS][MAX_LENGTHS])
{
    char path[400];
    mkdir("results", 0775);
    snprintf(path, sizeof path, "results/%s", code);
    mkdir(path, 0775);
    snprintf(path, sizeof path, "results/%s/%s.tex", code, corpus);
    FILE *fp = fopen(path, "w");
    if (!fp) return;
    printf("\tSaving data on %s/%s.tex\n", code, corpus);
    fprintf(fp, "\\begin{tabular}{|l|");
    for (int il = 0; o->lengths[il] > 0; ++il)
        if (o->lengths[il] >= o->minlen && o->lengths[il] <= o->maxlen) fprintf(fp, "l");
    fprintf(fp, "|}\n\\hline\n$m$");
    for (int il = 0; o->lengths[il] > 0; ++il)
        if (o->lengths[il] >= o->minlen && o->lengths[il] <= o->maxlen) fprintf(fp, " & $%d$", o->lengths[il]);
    fprintf(fp, "\\\\\n");
    for (int ia = 0; ia < o->nalgos; ++ia) {
        char name[32];
        upper(name, smartgpu_algo_name(o->algos[ia]));
        fprintf(fp, "\\textsc{%s}", name);
        for (int il = 0; o->lengths[il] > 0; ++il) {
            if (o->lengths[il] < o->minlen || o->lengths[il] > o->maxlen) continue;
            if (table[ia][il].mean > 0) fprintf(fp, " & %.2f", table[ia][il].mean);
            else fprintf(fp, " & -");
        }
        fprintf(fp, "\\\\\n");
    }
    fprintf(fp, "\\hline\n\\end{tabular}");
    fclose(fp);
}

/* results/<code>/<corpus>.xml (reference: outputXML, src/output.h:196-247): per algorithm
 * one <DATA><SEARCH>ms</SEARCH></DATA> per length, then the best time per length.  An
 * aborted cell is a single <DATA>-</DATA> (the reference prints that AND a 0.00 block);
 * <GPUS>, <HBMPEAK>, and per cell <GBS> (GB/s of text scanned), <ROOFLINE> (its share of GPUs x peak)
 * and <KERNEL> are this harness' additions. */
static void write_xml(const struct options *o, const char *corpus, const char *code,
                      struct cell table[MAX_ALGOS][MAX_LENGTHS])
{
    char path[400];
    mkdir("results", 0775);
    snprintf(path, sizeof path, "results/%s", code);
    mkdir(path, 0775);
    snprintf(path, sizeof path, "results/%s/%s.xml", code, corpus);
    FILE *fp = fopen(path, "w");
    if (!fp) return;
    printf("\tSaving data on %s/%s.xml\n", code, corpus);
    fprintf(fp, "<RESULTS>\n\t<CODE>%s</CODE>\n\t<TEXT>%s</TEXT>\n\t<GPUS>%d</GPUS>\n\t<HBMPEAK unit=\"GB/s per GPU\">%.0f</HBMPEAK>\n", code, corpus,
            o->gpus > 1 ? o->gpus : 1, HBM_PEAK_GBS);
    for (int ia = 0; ia < o->nalgos; ++ia) {
        char name[32];
        upper(name, smartgpu_algo_name(o->algos[ia]));
        fprintf(fp, "\t<ALGO>\n\t\t<NAME>%s</NAME>\n", name);
        for (int il = 0; o->lengths[il] > 0; ++il) {
            if (o->lengths[il] < o->minlen || o->lengths[il] > o->maxlen) continue;
            const struct cell *c = &table[ia][il];
            if (c->mean <= 0) { fprintf(fp, "\t\t<DATA>-</DATA>\n"); continue; }
            fprintf(fp, "\t\t<DATA>\n\t\t\t<SEARCH>%.4f</SEARCH>\n\t\t\t<GBS>%.1f</GBS>\n\t\t\t<ROOFLINE>%.4f</ROOFLINE>\n\t\t\t<KERNEL>%s</KERNEL>\n\t\t</DATA>\n",
                    c->mean, c->gbs, c->roof, c->kernel);
        }
        fprintf(fp, "\t</ALGO>\n");
    }
    fprintf(fp, "\t<BEST>\n");
    for (int il = 0; o->lengths[il] > 0; ++il) {
        if (o->lengths[il] < o->minlen || o->lengths[il] > o->maxlen) continue;
        double best = 999999.0;
        for (int ia = 0; ia < o->nalgos; ++ia)
            if (table[ia][il].mean > 0 && table[ia][il].mean < best) best = table[ia][il].mean;
        fprintf(fp, "\t\t<DATA>%.2f</DATA>\n", best);
    }
    fprintf(fp, "\t</BEST>\n</RESULTS>");
    fclose(fp);
}

/* results/<code>/<corpus>.php: the table as a PHP array for SMART's result browser (reference:
 * outputPHP, src/output.h:49-113): "PATT" => the lengths, "<ALGO>" => the mean times as strings
 * ("VOID" where there is none), with -dif "<ALGO>.best" / "<ALGO>.worst", with -std "<ALGO>.std".
 * Four decimals instead of the reference's two: a search of a 1 MiB text takes microseconds here. */
static void write_php(const struct options *o, const char *corpus, const char *code,
                      struct cell table[MAX_ALGOS][MAX_LENGTHS])
{
    char path[400];
    snprintf(path, sizeof path, "results/%s/%s.php", code, corpus);
    FILE *fp = fopen(path, "w");
    if (!fp) { printf("\tError in writing file %s/%s.php\n", code, corpus); return; }
    printf("\tSaving data on %s/%s.php\n", code, corpus);
    fprintf(fp, "<?\n$%s = array(\n\t\"PATT\" => array(", corpus);
    for (int il = 0; o->lengths[il] > 0; ++il)
        if (o->lengths[il] >= o->minlen && o->lengths[il] <= o->maxlen) fprintf(fp, "\"%d\", ", o->lengths[il]);
    fprintf(fp, "),\n");
    for (int ia = 0; ia < o->nalgos; ++ia) {
        char name[32];
        upper(name, smartgpu_algo_name(o->algos[ia]));
        for (int what = 0; what < 4; ++what) {  /* mean, best, worst, std */
            if ((what == 1 || what == 2) && !o->dif) continue;
            if (what == 3 && !o->std) continue;
            fprintf(fp, "\t\"%s%s\" => array(", name, what == 0 ? "" : what == 1 ? ".best" : what == 2 ? ".worst" : ".std");
            for (int il = 0; o->lengths[il] > 0; ++il) {
                if (o->lengths[il] < o->minlen || o->lengths[il] > o->maxlen) continue;
                const struct cell *c = &table[ia][il];
                const double v = what == 0 ? c->mean : what == 1 ? c->best : what == 2 ? c->worst : c->std;
                if (c->mean <= 0) fprintf(fp, (what == 1 || what == 2) ? "\"0.1\", " : "\"VOID\", ");
                else fprintf(fp, "\"%.4f\", ", v);
            }
            fprintf(fp, "),\n");
        }
    }
    fprintf(fp, ");\n?>");
    fclose(fp);
}

/* results/<code>/index.html: one link per corpus page (reference: outputINDEX, src/output.h:706-741) */
static void write_index(const char *code, const char *const *names, int ncorp)
{
    char path[400];
    snprintf(path, sizeof path, "results/%s/index.html", code);
    FILE *fp = fopen(path, "w");
    printf("\tWriting %s/index.html\n", code);
    if (!fp) { printf("\tError in writing file %s/index.html\n", code); return; }
    fprintf(fp, "<!DOCTYPE html><html><head><meta charset=\"utf-8\"><title>SMART: experimental results %s</title></head>\n<body>"
                "<h2>SMART: experimental results %s</h2>\n<table>\n", code, code);
    for (int k = 0; k < ncorp; ++k)
        fprintf(fp, "<tr><td><a href=\"%s.html\">Experimental results on %s</a></td></tr>\n", names[k], names[k]);
    fprintf(fp, "</table></body></html>\n");
    fclose(fp);
}

/* One line chart as inline SVG: a polyline per algorithm over the pattern lengths (equally spaced, as the
 * reference's charts place them), the value of cell(ia, il) on a linear axis from 0. */
static void svg_chart(FILE *fp, const struct options *o, struct cell table[MAX_ALGOS][MAX_LENGTHS], int gbs,
                      const char *title, const char *unit)
{
    static const char *const colour[] = {"#1f77b4", "#d62728", "#2ca02c", "#9467bd", "#ff7f0e", "#8c564b", "#e377c2", "#7f7f7f",
                                         "#bcbd22", "#17becf", "#393b79", "#ad494a", "#637939", "#7b4173", "#e6550d", "#3182bd"};
    int cols[MAX_LENGTHS], nc = 0;
    for (int il = 0; o->lengths[il] > 0; ++il)
        if (o->lengths[il] >= o->minlen && o->lengths[il] <= o->maxlen) cols[nc++] = il;
    double top = 0;
    for (int ia = 0; ia < o->nalgos; ++ia)
        for (int k = 0; k < nc; ++k) {
            const struct cell *c = &table[ia][cols[k]];
            const double v = c->mean > 0 ? (gbs ? c->gbs : c->mean) : 0;
            if (v > top) top = v;
        }
    if (nc == 0 || top <= 0) return;
    const int W = 760, H = 320, L = 60, R = 130, T = 30, B = 40;
    const double dx = nc > 1 ? (double)(W - L - R) / (nc - 1) : 0;
    fprintf(fp, "<h3>%s</h3>\n<svg xmlns=\"http://www.w3.org/2000/svg\" width=\"%d\" height=\"%d\" font-family=\"sans-serif\" font-size=\"11\">\n", title, W, H);
    fprintf(fp, "<rect x=\"%d\" y=\"%d\" width=\"%d\" height=\"%d\" fill=\"none\" stroke=\"#999\"/>\n", L, T, W - L - R, H - T - B);
    for (int g = 0; g <= 4; ++g) {  /* horizontal grid and the y axis' labels */
        const double y = T + (H - T - B) * (1.0 - g / 4.0);
        fprintf(fp, "<line x1=\"%d\" y1=\"%.1f\" x2=\"%d\" y2=\"%.1f\" stroke=\"#ddd\"/><text x=\"%d\" y=\"%.1f\" text-anchor=\"end\">%.4g</text>\n",
                L, y, W - R, y, L - 6, y + 4, top * g / 4.0);
    }
    for (int k = 0; k < nc; ++k)
        fprintf(fp, "<text x=\"%.1f\" y=\"%d\" text-anchor=\"middle\">%d</text>\n", L + dx * k, H - B + 16, o->lengths[cols[k]]);
    fprintf(fp, "<text x=\"%d\" y=\"%d\" text-anchor=\"middle\">pattern length</text><text x=\"12\" y=\"%d\">%s</text>\n",
            L + (W - L - R) / 2, H - 6, T - 10, unit);
    for (int ia = 0; ia < o->nalgos; ++ia) {
        char name[32];
        upper(name, smartgpu_algo_name(o->algos[ia]));
        const char *col = colour[ia % 16];
        fprintf(fp, "<polyline fill=\"none\" stroke=\"%s\" stroke-width=\"1.5\" points=\"", col);
        for (int k = 0; k < nc; ++k) {
            const struct cell *c = &table[ia][cols[k]];
            if (c->mean <= 0) continue;
            fprintf(fp, "%.1f,%.1f ", L + dx * k, T + (H - T - B) * (1.0 - (gbs ? c->gbs : c->mean) / top));
        }
        fprintf(fp, "\"/>\n<line x1=\"%d\" y1=\"%d\" x2=\"%d\" y2=\"%d\" stroke=\"%s\" stroke-width=\"2\"/><text x=\"%d\" y=\"%d\">%s</text>\n",
                W - R + 10, T + 8 + 14 * ia, W - R + 30, T + 8 + 14 * ia, col, W - R + 36, T + 12 + 14 * ia, name);
    }
    fprintf(fp, "</svg>\n");
}

/* results/<code>/<corpus>.html: the report page (reference: outputHTML2, src/output.h:443-633).
 * Same content — header block, one row per algorithm and one column per length, the best time
 * of a column in bold, the preprocessing time when -pre was given — as a self-contained page:
 * the reference's page draws its charts with the RGraph scripts and style sheet of its results/
 * directory, which are not part of this project; here the two charts (mean running time and GB/s
 * over the pattern lengths, a line per algorithm) are inline SVG, and the GB/s and the kernel of
 * each cell are listed under its time. */
static void write_html(const struct options *o, const char *corpus, const char *code, long long n,
                       struct cell table[MAX_ALGOS][MAX_LENGTHS])
{
    char path[400];
    mkdir("results", 0775);
    snprintf(path, sizeof path, "results/%s", code);
    mkdir(path, 0775);
    snprintf(path, sizeof path, "results/%s/%s.html", code, corpus);
    FILE *fp = fopen(path, "w");
    if (!fp) return;
    printf("\tSaving data on %s/%s.html\n", code, corpus);
    fprintf(fp, "<!DOCTYPE html><html><head><meta charset=\"utf-8\"><title>SMART Experimental Results %s: %s</title>\n", code, corpus);
    fprintf(fp, "<style>body{font-family:sans-serif}table{border-collapse:collapse}td{border:1px solid #999;padding:3px 8px;"
                "text-align:center}td.algo{text-align:left}.best{font-weight:bold}.pre,.gbs{font-size:70%%;color:#666}</style>"
                "</head><body>\n");
    fprintf(fp, "<h2>Report of Experimental Results</h2>\n<p>Test Code %s<br>Text %s (size : %lld bytes)<br>"
                "Engine: %s, %d GPU(s)</p>\n", code, corpus, n, smartgpu_version(), o->gpus > 1 ? o->gpus : 1);
    fprintf(fp, "<table id=\"resultTable\">\n<tr><td></td>");
    for (int il = 0; o->lengths[il] > 0; ++il)
        if (o->lengths[il] >= o->minlen && o->lengths[il] <= o->maxlen) fprintf(fp, "<td>%d</td>", o->lengths[il]);
    fprintf(fp, "</tr>\n");
    for (int ia = 0; ia < o->nalgos; ++ia) {
        char name[32];
        upper(name, smartgpu_algo_name(o->algos[ia]));
        fprintf(fp, "<tr><td class=\"algo\"><b>%s</b></td>", name);
        for (int il = 0; o->lengths[il] > 0; ++il) {
            if (o->lengths[il] < o->minlen || o->lengths[il] > o->maxlen) continue;
            const struct cell *c = &table[ia][il];
            double best = 0;
            for (int ib = 0; ib < o->nalgos; ++ib)
                if (table[ib][il].mean > 0 && (best == 0 || table[ib][il].mean < best)) best = table[ib][il].mean;
            fprintf(fp, "<td>");
            if (o->pre && c->mean > 0) fprintf(fp, "<div class=\"pre\">%.2f</div>", c->pre);
            if (c->mean <= 0) fprintf(fp, "<div>-</div>");
            else fprintf(fp, "<div%s>%.4f</div><div class=\"gbs\">%.1f GB/s<br>%s</div>", c->mean == best ? " class=\"best\"" : "", c->mean, c->gbs, c->kernel);
            fprintf(fp, "</td>");
        }
        fprintf(fp, "</tr>\n");
    }
    fprintf(fp, "</table>\n<p>Running times in milliseconds (mean over %d patterns)%s.</p>\n", o->runs,
            o->pre ? ", preprocessing times above them" : "");
    svg_chart(fp, o, table, 0, "Running times", "ms");
    svg_chart(fp, o, table, 1, "Text scanned per second", "GB/s");
    fprintf(fp, "</body></html>\n");
    fclose(fp);
}

int main(int argc, char **argv)
{
    struct options o;
    memset(&o, 0, sizeof o);
    o.runs = 500;
    o.tsize = 1048576;
    o.minlen = 1;
    o.maxlen = XSIZE;
    o.lengths = LEN_LARGE;
    o.limit_ms = 300;
    o.data_dir = "data";
    o.seed = -1;
    int custom_len[2] = {0, 0};
    /* dmabuf IPC only on this pool: RCCL between the GPUs of -gpus K needs it, and it is read when the runtime
     * initialises — so before the first HIP call of this process */
    setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);
    o.reduce = SMARTGPU_REDUCE_RCCL;
    { const char *e = getenv("SMARTGPU_REDUCE_HOST"); if (e && *e && *e != '0') o.reduce = SMARTGPU_REDUCE_HOST; }

    if (argc == 1) { printf("No parameter given. Use -h for help.\n\n"); return 0; }
    const char *bad = "Error in input parameters. Use -h for help.\n\n";
    for (int i = 1; i < argc; ++i) {
        const char *a = argv[i];
        int has1 = i + 1 < argc, has2 = i + 2 < argc;
        if (!strcmp(a, "-h")) { usage(); return 0; }
        else if (!strcmp(a, "-pset")) { if (!has1 || !is_number(argv[i + 1])) { printf("%s", bad); return 0; } o.runs = atoi(argv[++i]); }
        else if (!strcmp(a, "-tsize")) { if (!has1 || !is_number(argv[i + 1])) { printf("%s", bad); return 0; } o.tsize = atol(argv[++i]) * 1048576L; }
        else if (!strcmp(a, "-tb")) { if (!has1 || !is_decimal(argv[i + 1])) { printf("%s", bad); return 0; } o.limit_ms = atof(argv[++i]); o.each = 1; }
        else if (!strcmp(a, "-text")) { if (!has1) { printf("%s", bad); return 0; } snprintf(o.text_arg, sizeof o.text_arg, "%s", argv[++i]); }
        else if (!strcmp(a, "-plen")) {
            if (!has2 || !is_number(argv[i + 1]) || !is_number(argv[i + 2])) { printf("%s", bad); return 0; }
            o.minlen = atoi(argv[++i]);
            o.maxlen = atoi(argv[++i]);
            if (o.minlen < 1 || o.minlen > XSIZE) { printf("Error in input parameters. The minimum length is not a valid argument.\n\n"); return 0; }
            if (o.maxlen < 1 || o.minlen > o.maxlen) { printf("Error in input parameters. The maximum length is not a valid argument.\n\n"); return 0; }
        }
        else if (!strcmp(a, "-simple")) {
            if (!has2) { printf("%s", bad); return 0; }
            if (strlen(argv[i + 1]) > 100) { printf("Error in input parameters. Max 100 chars for P parameter.\n\n"); return 0; }
            if (strlen(argv[i + 2]) > 1000) { printf("Error in input parameters. Max 1000 chars for T parameter.\n\n"); return 0; }
            strcpy(o.simple_p, argv[++i]);
            strcpy(o.simple_t, argv[++i]);
            o.simple = 1;
        }
        else if (!strcmp(a, "-occ")) o.occ = 1;
        else if (!strcmp(a, "-pre")) o.pre = 1;
        else if (!strcmp(a, "-dif")) o.dif = o.each = 1;
        else if (!strcmp(a, "-std")) o.std = o.each = 1;
        else if (!strcmp(a, "-txt")) o.txt = 1;
        else if (!strcmp(a, "-tex")) o.tex = 1;
        else if (!strcmp(a, "-php")) o.php = 1;
        else if (!strcmp(a, "-short")) o.lengths = LEN_SHORT;
        else if (!strcmp(a, "-vshort")) o.lengths = LEN_VERY_SHORT;
        else if (!strcmp(a, "-data")) { if (!has1) { printf("%s", bad); return 0; } o.data_dir = argv[++i]; }
        else if (!strcmp(a, "-gpu")) { if (!has1 || !is_number(argv[i + 1])) { printf("%s", bad); return 0; } o.device = atoi(argv[++i]); }
        else if (!strcmp(a, "-seed")) { if (!has1 || !is_number(argv[i + 1])) { printf("%s", bad); return 0; } o.seed = atol(argv[++i]); }
        else if (!strcmp(a, "-gpus")) { if (!has1 || !is_number(argv[i + 1])) { printf("%s", bad); return 0; } o.gpus = atoi(argv[++i]); }
        else if (!strcmp(a, "-algo")) {
            if (!has1) { printf("%s", bad); return 0; }
            char list[256];
            snprintf(list, sizeof list, "%s", argv[++i]);
            for (char *tok = strtok(list, ","); tok; tok = strtok(NULL, ",")) {
                int id = smartgpu_algo_id(tok);
                if (id < 0) { printf("Error in input parameters. Unknown algorithm %s.\n\n", tok); return 0; }
                if (o.nalgos < MAX_ALGOS) o.algos[o.nalgos++] = id;
            }
        }
        else { printf("%s", bad); return 0; }
    }
    if (o.runs < 1 || o.runs > MAX_RUNS) { printf("%s", bad); return 0; }
    if (o.text_arg[0] && o.simple) { printf("Error in input parameters. Both parameters -simple and -text defined.\n\n"); return 0; }
    if (!o.text_arg[0] && !o.simple) { printf("Error in input parameters. No filename given.\n\n"); return 0; }
    if (o.nalgos == 0) {
        int n = read_registry("source/algorithms.h", o.algos);
        if (n > 0) o.nalgos = n;
        else for (int i = 0; i < MAX_ALGOS; ++i) o.algos[o.nalgos++] = i;
    }
    if (o.gpus > smartgpu_device_count() && o.reduce != SMARTGPU_REDUCE_HOST) {
        fprintf(stderr, "smart: -gpus %d but only %d GPU(s) visible\n", o.gpus, smartgpu_device_count());
        return 1;
    }
    if (smartgpu_device_count() <= o.device) {
        fprintf(stderr, "smart: no usable GPU %d: %s\n", o.device, smartgpu_last_error());
        return 1;
    }
    seed_draw(o.seed >= 0 ? (unsigned)o.seed : (unsigned)time(NULL));
    char code[64];
    snprintf(code, sizeof code, "EXP%d", (int)time(NULL));
    static struct cell table[MAX_ALGOS][MAX_LENGTHS];

    if (o.simple) { /* smart.c:570-596 */
        long n = (long)strlen(o.simple_t);
        int m = (int)strlen(o.simple_p);
        custom_len[0] = m;
        o.lengths = custom_len;
        printf("\n\tText of %ld chars : %s\n", n, o.simple_t);
        printf("\tPattern of %d chars : %s\n", m, o.simple_p);
        printf("\tStarting experimental tests with code %s\n", code);
        smartgpu_text *text = smartgpu_text_upload(o.simple_t, (uint64_t)n, o.device);
        if (!text) { fprintf(stderr, "smart: %s\n", smartgpu_last_error()); return 1; }
        run_corpus(&o, "", (const unsigned char *)o.simple_t, n, text, NULL, code, table);
        smartgpu_text_free(text);
        return 0;
    }

    /* corpus list: "all" or A-B-C (smart.c:597-667, split_filelsit function.h:112-129) */
    const char *names[MAX_CORPORA], *processed[MAX_CORPORA];
    int ncorp = 0, nproc = 0;
    char list[256];
    snprintf(list, sizeof list, "%s", o.text_arg);
    if (!strcmp(list, "all")) {
        for (int i = 0; i < MAX_CORPORA; ++i) names[ncorp++] = CORPUS[i];
    } else {
        for (char *tok = strtok(list, "-"); tok && ncorp < MAX_CORPORA; tok = strtok(NULL, "-")) names[ncorp++] = tok;
    }
    printf("\tStarting experimental tests with code %s\n", code);
    unsigned char *T = malloc((size_t)o.tsize + 16);
    for (int ic = 0; ic < ncorp; ++ic) {
        const char *corpus = names[ic];
        printf("\n\tTry to process archive (%d/%d) %s\n", ic + 1, ncorp, corpus);
        int sigma = sigma_of(corpus);
        if (!sigma) { printf("\tError in loading alphabet size\n"); continue; }
        smartgpu_text *text = NULL;
        long n = load_corpus(o.data_dir, corpus, T, o.tsize);
        if (n > 0) {
            T[n] = 0;
            text = smartgpu_text_upload(T, (uint64_t)n, o.device);
        } else if (!strncmp(corpus, "rand", 4)) {
            /* no corpus file: generate rand<sigma> on the device and read it
             * back once for pattern extraction */
            n = o.tsize;
            printf("\tNo index file under %s/%s: generating %ld bytes of rand%d on the device\n", o.data_dir, corpus, n, sigma);
            text = smartgpu_text_generate(0x5EED0001ull, sigma, 0, (uint64_t)n, o.device);
            if (text && smartgpu_text_read(text, 0, (uint64_t)n, T) != SMARTGPU_OK) { smartgpu_text_free(text); text = NULL; }
        } else {
            printf("\tError in loading text buffer. No index file exists.\n");
            continue;
        }
        if (!text) { fprintf(stderr, "smart: %s\n", smartgpu_last_error()); continue; }
        smartgpu_mtext *mtext = NULL;
        if (o.gpus > 1) {  /* the same bytes, sharded over GPUs 0..gpus-1 */
            int devs[64], *devlist = NULL;
            if (o.reduce == SMARTGPU_REDUCE_HOST && o.gpus <= 64) {  /* rehearsal: shard g on GPU g mod visible */
                const int vis = smartgpu_device_count();
                for (int g = 0; g < o.gpus; ++g) devs[g] = g % vis;
                devlist = devs;
            }
            mtext = smartgpu_mtext_upload(T, (uint64_t)n, o.gpus, devlist);
            if (!mtext) { fprintf(stderr, "smart: %s\n", smartgpu_last_error()); smartgpu_text_free(text); continue; }
            printf("\tText sharded over %d GPUs (%ld bytes each, %d bytes overlap), counts summed %s\n", o.gpus, n / o.gpus, SMARTGPU_XSIZE - 1,
                   o.reduce == SMARTGPU_REDUCE_HOST ? "on the host (SMARTGPU_REDUCE_HOST: rehearsal)" : "with one RCCL all-reduce");
        }
        alphabet_report(T, n);
        printf("\tText buffer of dimension %ld byte\n", n);
        time_t now = time(NULL);
        char stamp[32];
        strftime(stamp, sizeof stamp, "%Y:%m:%d %H:%M:%S", localtime(&now));
        printf("\tExperimental tests started on %s\n", stamp);
        memset(table, 0, sizeof table);
        run_corpus(&o, corpus, T, n, text, mtext, code, table);
        smartgpu_mtext_free(mtext);
        if (o.txt) write_txt(&o, corpus, code, table);
        write_xml(&o, corpus, code, table); /* always, as smart.c:388 */
        write_html(&o, corpus, code, (long long)n, table); /* always, as smart.c:389 */
        if (o.tex) write_tex(&o, corpus, code, table);
        if (o.php) write_php(&o, corpus, code, table);
        smartgpu_text_free(text);
        processed[nproc++] = corpus;
    }
    write_index(code, processed, nproc); /* smart.c:635,670; only the corpora that have a page */
    free(T);
    return 0;
}
