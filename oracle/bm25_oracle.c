/*
 * TEST INFRASTRUCTURE ONLY — see bm25_oracle.h.  Plain-C restatement of the reference hot path.
 *
 * Followed reference lines (read as text; /root/reference/...):
 *   tokenize / stop-words      include/textutil.hpp:13-37
 *   base-term filter           src/api_engine.cpp:388-397   (len<2 and stop-words dropped; order + duplicates kept)
 *   segment load               src/api_segment.cpp:45-136, include/barrels.hpp:12-71, include/indexio.hpp:13-29
 *   bm25_idf                   src/api_engine.cpp:45-47
 *   per-segment scoring loop   src/api_engine.cpp:441-482   (term order, fp32, count bounds the loop)
 *   candidate set / found      src/api_engine.cpp:485-495
 *   global top-K across segs   src/api_engine.cpp:434-435,499-504
 *
 * Differences that do not change results: a dense float[N] + touched[N] replaces the per-segment
 * unordered_map (same +0.0f start, same add order per doc); inverted files are read into memory
 * instead of 4-byte ifstream reads; ties are ordered canonically (score desc, seg asc, doc asc)
 * where the reference's order is an artefact of hash-table iteration.
 * Postings whose docId >= N (corrupt index; UB in the reference) are skipped.
 */
#define _GNU_SOURCE
#include "bm25_oracle.h"

#include <ctype.h>
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>

#define ORC_BARRELS 64u

typedef struct {
    char*    term;
    uint32_t order;      /* insertion order: unordered_map::emplace keeps the FIRST of duplicates */
    uint32_t term_id, df, count, barrel;
    uint64_t offset;
} lex_rec;

typedef struct {
    uint32_t  N;
    float     avgdl;
    uint32_t* doc_len;
    uint32_t  n_docs_file;
    lex_rec*  lex;
    uint32_t  n_lex;
    int       use_barrels;
    uint8_t*  inv[ORC_BARRELS];   /* inverted file contents (legacy: inv[0]) */
    uint64_t  inv_size[ORC_BARRELS];
    uint32_t  barrel_count;
} orc_segment;

struct orc_index {
    uint32_t     n_segs;
    orc_segment* segs;
};

static __thread char g_err[512];
const char* orc_error(void) { return g_err; }

/* ---------------------------------------------------------------- file helpers */
typedef struct { uint8_t* p; uint64_t n, pos; } rd;

static int slurp(const char* path, rd* r) {
    FILE* f = fopen(path, "rb");
    if (!f) return 0;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    r->p = (uint8_t*)malloc(n > 0 ? (size_t)n : 1);
    r->n = n > 0 ? (uint64_t)n : 0;
    r->pos = 0;
    if (r->n && fread(r->p, 1, r->n, f) != r->n) { fclose(f); free(r->p); return 0; }
    fclose(f);
    return 1;
}
static uint32_t rd_u32(rd* r) { uint32_t v = 0; if (r->pos + 4 <= r->n) { memcpy(&v, r->p + r->pos, 4); r->pos += 4; } else r->pos = r->n; return v; }
static uint64_t rd_u64(rd* r) { uint64_t v = 0; if (r->pos + 8 <= r->n) { memcpy(&v, r->p + r->pos, 8); r->pos += 8; } else r->pos = r->n; return v; }
static float rd_f32(rd* r) { float v = 0; if (r->pos + 4 <= r->n) { memcpy(&v, r->p + r->pos, 4); r->pos += 4; } else r->pos = r->n; return v; }
static char* rd_str(rd* r) {
    uint32_t n = rd_u32(r);
    if (r->pos + n > r->n) { r->pos = r->n; n = 0; }
    char* s = (char*)malloc((size_t)n + 1);
    memcpy(s, r->p + r->pos, n);
    s[n] = 0;
    r->pos += n;
    return s;
}
static int exists(const char* p) { struct stat st; return stat(p, &st) == 0; }

static int lex_cmp(const void* a, const void* b) {
    const lex_rec* x = (const lex_rec*)a; const lex_rec* y = (const lex_rec*)b;
    int c = strcmp(x->term, y->term);
    if (c) return c;
    return x->order < y->order ? -1 : (x->order > y->order ? 1 : 0);
}

static void read_lex(rd* r, uint32_t barrel, orc_segment* s, uint32_t* cap) {
    uint32_t tcount = rd_u32(r);
    for (uint32_t i = 0; i < tcount; i++) {
        if (s->n_lex == *cap) { *cap = *cap ? *cap * 2 : 1024; s->lex = (lex_rec*)realloc(s->lex, (size_t)*cap * sizeof(lex_rec)); }
        lex_rec* e = &s->lex[s->n_lex];
        e->term = rd_str(r);
        e->term_id = rd_u32(r);
        e->df = rd_u32(r);
        e->offset = rd_u64(r);
        e->count = rd_u32(r);
        e->barrel = barrel;
        e->order = s->n_lex;
        s->n_lex++;
    }
}

static int load_segment(const char* dir, orc_segment* s) {
    char path[4096];
    rd r;
    memset(s, 0, sizeof(*s));
    snprintf(path, sizeof(path), "%s/stats.bin", dir);
    if (!slurp(path, &r)) return 0;
    s->N = rd_u32(&r); s->avgdl = rd_f32(&r); free(r.p);
    snprintf(path, sizeof(path), "%s/docs.bin", dir);
    if (!slurp(path, &r)) return 0;
    uint32_t n = rd_u32(&r);
    s->n_docs_file = n;
    uint32_t alloc = n > s->N ? n : s->N;
    s->doc_len = (uint32_t*)calloc(alloc ? alloc : 1, 4);
    for (uint32_t i = 0; i < n; i++) {
        free(rd_str(&r)); free(rd_str(&r)); free(rd_str(&r));
        s->doc_len[i] = rd_u32(&r);
    }
    free(r.p);
    uint32_t cap = 0;
    char p0[4096], p1[4096];
    snprintf(path, sizeof(path), "%s/barrels.bin", dir);
    snprintf(p0, sizeof(p0), "%s/inverted_b000.bin", dir);
    snprintf(p1, sizeof(p1), "%s/lexicon_b000.bin", dir);
    if (exists(path) && exists(p0) && exists(p1)) {   /* has_barrels, include/barrels.hpp:67-71 */
        s->use_barrels = 1;
        if (!slurp(path, &r)) return 0;
        s->barrel_count = rd_u32(&r); (void)rd_u32(&r); free(r.p);
        if (s->barrel_count > ORC_BARRELS) { snprintf(g_err, sizeof(g_err), "barrel_count %u > %u", s->barrel_count, ORC_BARRELS); return 0; }
        for (uint32_t b = 0; b < s->barrel_count; b++) {
            snprintf(path, sizeof(path), "%s/inverted_b%03u.bin", dir, b);
            rd iv;
            if (!slurp(path, &iv)) return 0;
            s->inv[b] = iv.p; s->inv_size[b] = iv.n;
        }
        for (uint32_t b = 0; b < s->barrel_count; b++) {
            snprintf(path, sizeof(path), "%s/lexicon_b%03u.bin", dir, b);
            if (!slurp(path, &r)) return 0;
            read_lex(&r, b, s, &cap);
            free(r.p);
        }
    } else {
        s->use_barrels = 0;
        s->barrel_count = 1;
        snprintf(path, sizeof(path), "%s/lexicon.bin", dir);
        if (!slurp(path, &r)) return 0;
        read_lex(&r, 0, s, &cap);
        free(r.p);
        snprintf(path, sizeof(path), "%s/inverted.bin", dir);
        rd iv;
        if (!slurp(path, &iv)) return 0;
        s->inv[0] = iv.p; s->inv_size[0] = iv.n;
    }
    if (s->n_lex) qsort(s->lex, s->n_lex, sizeof(lex_rec), lex_cmp);
    return 1;
}

static const lex_rec* lex_find(const orc_segment* s, const char* term) {
    uint32_t lo = 0, hi = s->n_lex;
    while (lo < hi) {   /* first record with term >= key: duplicates sort by insertion order, so this is the first inserted */
        uint32_t mid = lo + (hi - lo) / 2;
        if (strcmp(s->lex[mid].term, term) < 0) lo = mid + 1; else hi = mid;
    }
    if (lo < s->n_lex && strcmp(s->lex[lo].term, term) == 0) return &s->lex[lo];
    return NULL;
}

orc_index* orc_open(const char* index_dir) {
    char path[4096];
    g_err[0] = 0;
    snprintf(path, sizeof(path), "%s/manifest.bin", index_dir);
    rd r;
    if (!slurp(path, &r)) { snprintf(g_err, sizeof(g_err), "cannot read %s", path); return NULL; }
    uint32_t n = rd_u32(&r);
    orc_index* ix = (orc_index*)calloc(1, sizeof(*ix));
    ix->segs = (orc_segment*)calloc(n ? n : 1, sizeof(orc_segment));
    for (uint32_t i = 0; i < n; i++) {
        char* name = rd_str(&r);
        snprintf(path, sizeof(path), "%s/segments/%s", index_dir, name);
        free(name);
        if (!load_segment(path, &ix->segs[i])) {
            if (!g_err[0]) snprintf(g_err, sizeof(g_err), "failed to load segment %s", path);
            ix->n_segs = i + 1;
            free(r.p);
            orc_close(ix);
            return NULL;
        }
        ix->n_segs = i + 1;
    }
    free(r.p);
    return ix;
}

void orc_close(orc_index* ix) {
    if (!ix) return;
    for (uint32_t i = 0; i < ix->n_segs; i++) {
        orc_segment* s = &ix->segs[i];
        for (uint32_t j = 0; j < s->n_lex; j++) free(s->lex[j].term);
        free(s->lex);
        free(s->doc_len);
        for (uint32_t b = 0; b < ORC_BARRELS; b++) free(s->inv[b]);
    }
    free(ix->segs);
    free(ix);
}

uint32_t orc_num_segments(const orc_index* ix) { return ix ? ix->n_segs : 0; }
uint32_t orc_segment_docs(const orc_index* ix, uint32_t seg) { return (ix && seg < ix->n_segs) ? ix->segs[seg].N : 0; }

/* ---------------------------------------------------------------- query text */
static int is_stop(const char* t) {
    static const char* const sw[] = {"the","a","an","and","or","of","to","in","for","on","with","by","as",
                                     "is","are","was","were","be","been","it","this","that","from","at"};
    for (size_t i = 0; i < sizeof(sw) / sizeof(sw[0]); i++) if (strcmp(t, sw[i]) == 0) return 1;
    return 0;
}

/* returns malloc'd array of malloc'd terms */
static char** query_terms(const char* q, uint32_t* n_out) {
    size_t len = strlen(q);
    char** out = (char**)malloc((len / 2 + 2) * sizeof(char*));
    uint32_t n = 0;
    char* cur = (char*)malloc(len + 1);
    size_t cl = 0;
    for (size_t i = 0; i <= len; i++) {
        unsigned char uc = (unsigned char)q[i];
        if (i < len && isalnum(uc)) {
            cur[cl++] = (char)tolower(uc);
        } else if (cl) {
            cur[cl] = 0;
            if (cl >= 2 && !is_stop(cur)) out[n++] = strdup(cur);
            cl = 0;
        }
    }
    free(cur);
    *n_out = n;
    return out;
}
static void free_terms(char** t, uint32_t n) { for (uint32_t i = 0; i < n; i++) free(t[i]); free(t); }

static float bm25_idf(uint32_t N, uint32_t df) {
    return logf((((N - df + 0.5f) / (df + 0.5f)) + 1.0f));
}

/* ---------------------------------------------------------------- scoring */
typedef struct {
    float*    acc;       /* max N over segments */
    uint8_t*  touched;
    uint16_t* mcount;    /* AND: term refs that hit the doc */
    uint32_t* tlist;     /* touched docIds, for O(found) reset */
    uint32_t  cap;
} scratch;

static void scratch_init(scratch* s, const orc_index* ix) {
    uint32_t mx = 1;
    for (uint32_t i = 0; i < ix->n_segs; i++) if (ix->segs[i].N > mx) mx = ix->segs[i].N;
    s->cap = mx;
    s->acc = (float*)calloc(mx, sizeof(float));
    s->touched = (uint8_t*)calloc(mx, 1);
    s->mcount = (uint16_t*)calloc(mx, sizeof(uint16_t));
    s->tlist = (uint32_t*)malloc((size_t)mx * sizeof(uint32_t));
}
static void scratch_free(scratch* s) { free(s->acc); free(s->touched); free(s->mcount); free(s->tlist); }

/* Scores one segment for the given terms into sc->acc/touched; returns number of touched docs
 * (n_touched) and the number of scored term refs (n_refs).  Leaves state for the caller to read,
 * caller must call seg_reset afterwards. */
static uint32_t score_segment(const orc_segment* seg, char** terms, uint32_t nterms, scratch* sc, uint32_t* n_refs_out, uint64_t* postings_out) {
    const float k1 = 1.2f;
    const float b = 0.75f;
    uint32_t nt = 0, n_refs = 0;
    uint64_t npost = 0;
    for (uint32_t t = 0; t < nterms; t++) {
        const float qweight = 1.0f;                       /* src/api_engine.cpp:419-421 (no embeddings) */
        const lex_rec* e = lex_find(seg, terms[t]);       /* :454-455 */
        if (!e) continue;
        if (e->df == 0) continue;                          /* :458 */
        float idf = bm25_idf(seg->N, e->df);               /* :461 */
        uint32_t bsel = seg->use_barrels ? e->barrel : 0;  /* :464-466 */
        const uint8_t* base = seg->inv[bsel];
        uint64_t size = seg->inv_size[bsel];
        n_refs++;
        for (uint32_t i = 0; i < e->count; i++) {          /* :473 */
            uint64_t off = e->offset + (uint64_t)i * 8;
            uint32_t docId = 0, tf = 0;
            if (off + 8 <= size) { memcpy(&docId, base + off, 4); memcpy(&tf, base + off + 4, 4); }
            else break;                                     /* past EOF: the reference would read garbage */
            npost++;
            if (docId >= seg->N) continue;                  /* corrupt posting: UB in the reference */
            float dl = (float)seg->doc_len[docId];
            float denom = (float)tf + k1 * (1.0f - b + b * (dl / seg->avgdl));   /* :478 */
            float s = idf * ((float)tf * (k1 + 1.0f)) / denom;                  /* :479 */
            if (!sc->touched[docId]) { sc->touched[docId] = 1; sc->acc[docId] = 0.0f; sc->mcount[docId] = 0; sc->tlist[nt++] = docId; }
            sc->acc[docId] += qweight * s;                                       /* :480 */
            sc->mcount[docId]++;
        }
    }
    *n_refs_out = n_refs;
    if (postings_out) *postings_out += npost;
    return nt;
}
static void seg_reset(scratch* sc, uint32_t nt) { for (uint32_t i = 0; i < nt; i++) sc->touched[sc->tlist[i]] = 0; }

/* canonical order: a is better than b */
static int better(const orc_hit* a, const orc_hit* b) {
    if (a->score > b->score) return 1;
    if (a->score < b->score) return 0;
    if (a->seg != b->seg) return a->seg < b->seg;
    return a->doc < b->doc;
}

/* bounded min-heap (worst at root) of the K best hits */
static void heap_sift_down(orc_hit* h, uint32_t n, uint32_t i) {
    for (;;) {
        uint32_t l = 2 * i + 1, r = l + 1, w = i;
        if (l < n && better(&h[w], &h[l])) w = l;
        if (r < n && better(&h[w], &h[r])) w = r;
        if (w == i) return;
        orc_hit t = h[i]; h[i] = h[w]; h[w] = t;
        i = w;
    }
}
static void heap_offer(orc_hit* h, uint32_t* n, uint32_t K, orc_hit x) {
    if (*n < K) {
        uint32_t i = (*n)++;
        h[i] = x;
        while (i > 0) {
            uint32_t p = (i - 1) / 2;
            if (better(&h[p], &h[i])) { orc_hit t = h[p]; h[p] = h[i]; h[i] = t; i = p; } else break;
        }
    } else if (better(&x, &h[0])) {
        h[0] = x;
        heap_sift_down(h, *n, 0);
    }
}

static int search_one(orc_index* ix, scratch* sc, const char* query, int k, uint32_t flags, orc_hit* hits, uint32_t* nhits, uint64_t* found) {
    const uint32_t K = (uint32_t)(k < 1 ? 1 : (k > 100 ? 100 : k));      /* :377 */
    uint32_t nterms = 0;
    char** terms = query_terms(query, &nterms);
    *nhits = 0; *found = 0;
    if (nterms == 0 || ix->n_segs == 0) { free_terms(terms, nterms); return 0; }   /* :407 */
    orc_hit heap[100];
    uint32_t hn = 0;
    uint64_t total = 0;
    for (uint32_t sid = 0; sid < ix->n_segs; sid++) {                       /* :441 */
        const orc_segment* seg = &ix->segs[sid];
        uint32_t n_refs = 0;
        uint32_t nt = score_segment(seg, terms, nterms, sc, &n_refs, NULL);
        uint32_t cand = 0;
        for (uint32_t i = 0; i < nt; i++) {                                 /* :485-492, canonical instead of hash order */
            uint32_t d = sc->tlist[i];
            if ((flags & ORC_FLAG_AND) && sc->mcount[d] != n_refs) continue;
            cand++;
            orc_hit x; x.score = sc->acc[d]; x.seg = sid; x.doc = d;
            heap_offer(heap, &hn, K, x);
        }
        total += cand;                                                      /* :495 */
        seg_reset(sc, nt);
    }
    /* drain worst-first, then reverse (:499-504) */
    uint32_t n = hn;
    for (uint32_t i = n; i > 0; i--) {
        hits[i - 1] = heap[0];
        heap[0] = heap[--hn];
        heap_sift_down(heap, hn, 0);
    }
    *nhits = n;
    *found = total;
    free_terms(terms, nterms);
    return 1;
}

int orc_search(orc_index* ix, const char* query, int k, uint32_t flags, orc_hit* hits, uint32_t* nhits, uint64_t* found) {
    if (!ix || !query) return -1;
    scratch sc;
    scratch_init(&sc, ix);
    int rc = search_one(ix, &sc, query, k, flags, hits, nhits, found);
    scratch_free(&sc);
    return rc;
}

typedef struct {
    orc_index* ix; const char* const* queries; uint32_t begin, end; int k; uint32_t flags;
    orc_hit* hits; uint32_t* nhits; uint64_t* found; uint8_t* usable;
} job;

static void* worker(void* p) {
    job* j = (job*)p;
    const uint32_t K = (uint32_t)(j->k < 1 ? 1 : (j->k > 100 ? 100 : j->k));
    scratch sc;
    scratch_init(&sc, j->ix);
    for (uint32_t q = j->begin; q < j->end; q++) {
        int rc = search_one(j->ix, &sc, j->queries[q], j->k, j->flags, j->hits + (size_t)q * K, &j->nhits[q], &j->found[q]);
        if (j->usable) j->usable[q] = rc == 1;
    }
    scratch_free(&sc);
    return NULL;
}

int orc_search_batch(orc_index* ix, const char* const* queries, uint32_t n_queries, int k, uint32_t flags,
                     orc_hit* hits, uint32_t* nhits, uint64_t* found, uint8_t* usable, int threads) {
    if (!ix) return -1;
    if (threads < 1) threads = 1;
    if ((uint32_t)threads > n_queries) threads = n_queries ? (int)n_queries : 1;
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)threads);
    job* jobs = (job*)malloc(sizeof(job) * (size_t)threads);
    uint32_t per = (n_queries + (uint32_t)threads - 1) / (uint32_t)threads;
    for (int t = 0; t < threads; t++) {
        uint32_t b = (uint32_t)t * per, e = b + per;
        if (b > n_queries) b = n_queries;
        if (e > n_queries) e = n_queries;
        job jb = {ix, queries, b, e, k, flags, hits, nhits, found, usable};
        jobs[t] = jb;
        if (threads == 1) worker(&jobs[t]); else pthread_create(&th[t], NULL, worker, &jobs[t]);
    }
    if (threads > 1) for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
    free(th); free(jobs);
    return 0;
}

int orc_scores(orc_index* ix, const char* query, uint32_t flags, uint32_t seg, float* acc, uint8_t* touched) {
    if (!ix || seg >= ix->n_segs) return -1;
    uint32_t nterms = 0;
    char** terms = query_terms(query, &nterms);
    const orc_segment* s = &ix->segs[seg];
    memset(touched, 0, s->N);
    for (uint32_t i = 0; i < s->N; i++) acc[i] = 0.0f;
    if (nterms == 0) { free_terms(terms, nterms); return 0; }
    scratch sc;
    scratch_init(&sc, ix);
    uint32_t n_refs = 0;
    uint32_t nt = score_segment(s, terms, nterms, &sc, &n_refs, NULL);
    for (uint32_t i = 0; i < nt; i++) {
        uint32_t d = sc.tlist[i];
        if ((flags & ORC_FLAG_AND) && sc.mcount[d] != n_refs) continue;
        acc[d] = sc.acc[d];
        touched[d] = 1;
    }
    scratch_free(&sc);
    free_terms(terms, nterms);
    return 1;
}

uint64_t orc_query_postings(orc_index* ix, const char* query) {
    if (!ix) return 0;
    uint32_t nterms = 0;
    char** terms = query_terms(query, &nterms);
    uint64_t total = 0;
    for (uint32_t sid = 0; sid < ix->n_segs; sid++)
        for (uint32_t t = 0; t < nterms; t++) {
            const lex_rec* e = lex_find(&ix->segs[sid], terms[t]);
            if (e && e->df) total += e->count;
        }
    free_terms(terms, nterms);
    return total;
}
