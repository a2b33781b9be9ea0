/*
 * textgen — SMART's random corpora, byte for byte, without depending on the C library's rand().
 *
 * The reference tool (src/textgen.c:34-54) writes eight 5,000,000-byte files,
 * data/rand<s>/rand<s>.txt for s = 2,4,8,16,32,64,128,250, from ONE unseeded glibc rand()
 * stream: byte = rand() % s, the files in that order.  This tool restates that generator
 * (glibc's TYPE_3 additive feedback: 31 words seeded by the Lehmer sequence 16807*x mod 2^31-1,
 * the first 310 outputs discarded, output = (st[f] += st[b]) >> 1), so the corpora — and
 * their md5 sums listed in SURVEY.md §8c — are reproducible on any libc.  It also writes
 * the index.txt each corpus directory needs (getText, src/smart.c:95-138), when absent.
 *
 *   textgen [-data DIR]          default DIR = data
 *   textgen -sum                 print a 64-bit FNV-1a of every corpus instead of writing files
 */
#include <errno.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <sys/stat.h>

#define CORPUS_BYTES 5000000u

static uint32_t ring[31];
static int front, back;

static void stream_start(void)
{
    int32_t x = 1;
    ring[0] = 1;
    for (int i = 1; i < 31; ++i) {
        /* x = 16807 * x mod (2^31 - 1) without overflow: 2^31-1 = 16807*127773 + 2836 */
        x = 16807 * (x % 127773) - 2836 * (x / 127773);
        if (x < 0) x += 2147483647;
        ring[i] = (uint32_t)x;
    }
    front = 3;
    back = 0;
}

static uint32_t stream_next(void)
{
    ring[front] += ring[back];
    const uint32_t r = ring[front] >> 1;
    if (++front == 31) front = 0;
    if (++back == 31) back = 0;
    return r;
}

static int make_dir(const char *path)
{
    return (mkdir(path, 0777) == 0 || errno == EEXIST) ? 0 : -1;
}

int main(int argc, char **argv)
{
    static const int sigmas[8] = {2, 4, 8, 16, 32, 64, 128, 250};
    static unsigned char buf[CORPUS_BYTES];
    const char *data = "data";
    int sum_only = 0;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "-data") && i + 1 < argc) data = argv[++i];
        else if (!strcmp(argv[i], "-sum")) sum_only = 1;
        else {
            printf("usage: textgen [-data DIR] [-sum]\n");
            return 1;
        }
    }
    stream_start();
    for (int i = 0; i < 310; ++i) (void)stream_next();
    if (!sum_only && make_dir(data) != 0) { perror(data); return 1; }
    for (int k = 0; k < 8; ++k) {
        const uint32_t sigma = (uint32_t)sigmas[k];
        for (uint32_t i = 0; i < CORPUS_BYTES; ++i) buf[i] = (unsigned char)(stream_next() % sigma);
        if (sum_only) {
            uint64_t h = 1469598103934665603ull;
            for (uint32_t i = 0; i < CORPUS_BYTES; ++i) h = (h ^ buf[i]) * 1099511628211ull;
            printf("rand%d %016llx\n", sigmas[k], (unsigned long long)h);
            continue;
        }
        char dir[600], path[700];
        snprintf(dir, sizeof dir, "%s/rand%d", data, sigmas[k]);
        if (make_dir(dir) != 0) { perror(dir); return 1; }
        snprintf(path, sizeof path, "%s/rand%d.txt", dir, sigmas[k]);
        FILE *f = fopen(path, "wb");
        if (!f || fwrite(buf, 1, CORPUS_BYTES, f) != CORPUS_BYTES || fclose(f) != 0) { perror(path); return 1; }
        snprintf(path, sizeof path, "%s/index.txt", dir);
        f = fopen(path, "r");
        if (f) {
            fclose(f);
        } else if ((f = fopen(path, "w")) != NULL) {
            fprintf(f, "RANDOM TEXT OVER AN ALPHABET OF SIZE %d\nTOTAL SIZE: 5,0 MB\n\n#rand%d.txt#\n5,0 MB\n", sigmas[k], sigmas[k]);
            fclose(f);
        }
        printf("\trand%-3d  %s/rand%d.txt  %u bytes\n", sigmas[k], dir, sigmas[k], CORPUS_BYTES);
    }
    return 0;
}
