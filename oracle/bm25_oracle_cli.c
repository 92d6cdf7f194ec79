/* TEST INFRASTRUCTURE ONLY — CLI over bm25_oracle.c with the same output format as ref_driver:
 *   bm25_oracle_cli search <index_dir> <queries.txt> <K> <out.txt> [and]
 *   bm25_oracle_cli time   <index_dir> <queries.txt> <K> <max_seconds> [threads]
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "bm25_oracle.h"

static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; }

int main(int argc, char** argv) {
    if (argc < 6) { fprintf(stderr, "usage: %s search|time <index_dir> <queries.txt> <K> <out|max_seconds> [and|threads]\n", argv[0]); return 2; }
    orc_index* ix = orc_open(argv[2]);
    if (!ix) { fprintf(stderr, "orc_open: %s\n", orc_error()); return 1; }
    FILE* qf = fopen(argv[3], "r");
    if (!qf) { perror("queries"); return 1; }
    int K = atoi(argv[4]);
    char** qs = NULL; size_t nq = 0, cap = 0;
    char* line = NULL; size_t ln = 0; ssize_t got;
    while ((got = getline(&line, &ln, qf)) >= 0) {
        if (got && line[got - 1] == '\n') line[got - 1] = 0;
        if (nq == cap) { cap = cap ? cap * 2 : 256; qs = (char**)realloc(qs, cap * sizeof(char*)); }
        qs[nq++] = strdup(line);
    }
    fclose(qf);
    if (strcmp(argv[1], "search") == 0) {
        uint32_t flags = (argc > 6 && strcmp(argv[6], "and") == 0) ? ORC_FLAG_AND : ORC_FLAG_OR;
        FILE* out = fopen(argv[5], "w");
        if (!out) { perror("out"); return 1; }
        orc_hit hits[100];
        for (size_t i = 0; i < nq; i++) {
            uint32_t n = 0; uint64_t found = 0;
            int rc = orc_search(ix, qs[i], K, flags, hits, &n, &found);
            fprintf(out, "Q %lld %u\n", rc == 1 ? (long long)found : -1LL, n);
            for (uint32_t j = 0; j < n; j++) { uint32_t bits; memcpy(&bits, &hits[j].score, 4); fprintf(out, "%u %u %08x\n", hits[j].seg, hits[j].doc, bits); }
        }
        fclose(out);
    } else {
        double max_s = atof(argv[5]);
        int threads = argc > 6 ? atoi(argv[6]) : 1;
        uint32_t Kc = (uint32_t)(K < 1 ? 1 : (K > 100 ? 100 : K));
        size_t chunk = 64 * (size_t)threads, done = 0;
        orc_hit* hits = (orc_hit*)malloc(sizeof(orc_hit) * Kc * chunk);
        uint32_t* nh = (uint32_t*)malloc(4 * chunk); uint64_t* fd = (uint64_t*)malloc(8 * chunk);
        double t0 = now_s();
        while (done < nq && now_s() - t0 < max_s) {
            size_t n = nq - done < chunk ? nq - done : chunk;
            orc_search_batch(ix, (const char* const*)(qs + done), (uint32_t)n, K, 0, hits, nh, fd, NULL, threads);
            done += n;
        }
        double el = now_s() - t0;
        printf("{\"queries\": %zu, \"seconds\": %.6f, \"qps\": %.6f, \"threads\": %d}\n", done, el, done / el, threads);
    }
    orc_close(ix);
    return 0;
}
