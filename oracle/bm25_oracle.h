/*
 * TEST INFRASTRUCTURE ONLY — CPU oracle for the BM25 posting-traversal hot path.
 *
 * A plain-C restatement of the reference's algorithm (NOT a copy of its code; different data
 * structures, same arithmetic), used only by tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg as the CHECKER.  Nothing under nextsearch-api_amd/ links, imports or calls it.
 *
 * Parity status: PINNED — checked against golden vectors captured from the real reference engine
 * (oracle/_ref/ref_driver, built from /root/reference in place) in tests/test_oracle_golden.py;
 * fixtures under tests/golden/, generator tools/gen_golden.py.
 */
#ifndef BM25_ORACLE_H
#define BM25_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_FLAG_OR  0u
#define ORC_FLAG_AND 1u   /* derived extension: keep docs hit by every scored term ref of their segment */

typedef struct orc_index orc_index;

typedef struct orc_hit {   /* struct Hit, src/api_engine.cpp:427-431 */
    float    score;
    uint32_t seg;
    uint32_t doc;
} orc_hit;

orc_index*  orc_open(const char* index_dir);   /* Engine::reload + load_segment: src/api_engine.cpp:50-90, src/api_segment.cpp:105-136 */
void        orc_close(orc_index* ix);
const char* orc_error(void);
uint32_t    orc_num_segments(const orc_index* ix);
uint32_t    orc_segment_docs(const orc_index* ix, uint32_t seg);

/* Engine::search restated (src/api_engine.cpp:369-505).  K = clamp(k,1,100).  hits: K entries.
 * Order: score desc, then seg asc, then doc asc (the canonical choice inside the reference's
 * unspecified tie order).  Returns 1 when the "found" key would be present, 0 on the early-return
 * path (no usable terms / no segments, :407), <0 on error. */
int orc_search(orc_index* ix, const char* query, int k, uint32_t flags, orc_hit* hits, uint32_t* nhits, uint64_t* found);

/* Same for a batch, on `threads` host threads (queries partitioned, one accumulator set each). */
int orc_search_batch(orc_index* ix, const char* const* queries, uint32_t n_queries, int k, uint32_t flags,
                     orc_hit* hits, uint32_t* nhits, uint64_t* found, uint8_t* usable, int threads);

/* Dense per-segment accumulators of one query, for the tie-aware comparator: acc[N] (fp32 sums in
 * query-term order, src/api_engine.cpp:480) and touched[N] (1 = doc is a candidate). */
int orc_scores(orc_index* ix, const char* query, uint32_t flags, uint32_t seg, float* acc, uint8_t* touched);

/* Posting count the query touches (sum of LexEntry.count over scored term refs): algorithmic bytes / 8. */
uint64_t orc_query_postings(orc_index* ix, const char* query);

#ifdef __cplusplus
}
#endif
#endif
