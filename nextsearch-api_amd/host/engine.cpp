#include "engine.hpp"

#include <algorithm>
#include <memory>
#include <mutex>
#include <thread>
#include <cmath>
#include <limits>
#include <cstring>

#include "json_writer.hpp"
#include "textutil.hpp"

namespace nextsearch {

float bm25_idf(uint32_t N, uint32_t df) {
    // std::log((((N - df + 0.5f) / (df + 0.5f)) + 1.0f))   src/api_engine.cpp:45-47
    float num = (float)(uint32_t)(N - df) + 0.5f;
    float den = (float)df + 0.5f;
    return std::log((num / den) + 1.0f);   // float overload == glibc logf
}

Engine::Engine(int device) : device_(device) {}

Engine::Engine(const std::vector<int>& devices) : device_(devices.empty() ? 0 : devices[0]) {
    for (size_t i = 1; i < devices.size(); i++) { Replica r; r.device = devices[i]; replicas_.push_back(r); }
}

Engine::~Engine() {
    release_device_segments();
    if (ctx_) ns_ctx_destroy(ctx_);
    for (auto& r : replicas_) if (r.ctx) ns_ctx_destroy(r.ctx);   // frees the segments it holds
}

void Engine::release_device_segments() {
    if (ctx_)
        for (ns_seg* s : dev_segs_)
            if (s) ns_segment_release(ctx_, s);
    dev_segs_.clear();
    if (ctx_ && sem.dev) ns_sem_release(ctx_, sem.dev);
    sem.dev = nullptr;
}

// One segment's postings to the device without a host copy: every inverted file is mapped, appended from the mapping
// (the C-ABI stages it through pinned memory) and unmapped again.  What the reference's open ifstreams are to it
// (src/api_segment.cpp:70-102) the device buffer is to this engine.
static bool upload_segment(ns_ctx* ctx, uint32_t seg_id, nsx::SegmentData& s, ns_seg** out, std::string& err) {
    if (s.doc_len.size() < s.N) s.doc_len.resize(s.N, 0);   // stats.bin N larger than docs.bin: treat missing as 0
    ns_seg* dev = nullptr;
    int rc = ns_segment_upload_begin(ctx, seg_id, s.N, s.avgdl, s.doc_len.data(), s.postings_bytes, &dev);
    if (rc != NS_OK) { err = std::string("ns_segment_upload: ") + ns_last_error(ctx); return false; }
    const bool fed = nsx::for_each_inverted_file(s, [&](const uint8_t* p, uint64_t n) {
        rc = ns_segment_upload_append(ctx, dev, p, n);
        return rc == NS_OK;
    });
    if (fed) rc = ns_segment_upload_end(ctx, dev);
    if (!fed || rc != NS_OK) {
        err = rc != NS_OK ? std::string("ns_segment_upload: ") + ns_last_error(ctx)
                          : "cannot map the inverted files of " + s.dir.string() + " (missing, or changed since they were listed)";
        (void)ns_segment_release(ctx, dev);
        return false;
    }
    // skip tables for the segment's frequent lists (ns_segment_build_skips: 4 B per list and 1024-doc cell).  The
    // reference's lists carry no block metadata (src/lexicon.cpp:104-128); this is built from the uploaded postings.
    // The tables are an accelerator, never a load precondition: the reference loads a segment with a damaged lexicon
    // record and only the queries naming that term go wrong (src/api_engine.cpp:464-476), so records that do not
    // describe a whole list inside the payload are left out here, and a refusal of the rest costs the tables, not the
    // segment (queries then take the cursor path; a term ref of a damaged record is rejected at prepare, as before).
    {
        const uint32_t min_count = std::max<uint32_t>(64u, s.N / 512u);
        std::vector<uint64_t> off;
        std::vector<uint32_t> cnt;
        for (const auto& kv : s.lex) {
            const nsx::LexEntry& e = kv.second;
            if (e.df == 0 || e.count < min_count) continue;
            if (s.use_barrels && e.barrelId >= s.barrel_base.size()) continue;
            const uint64_t bo = s.list_byte_offset(e);
            if (bo % 8 != 0 || bo / 8 + e.count > s.postings_bytes / 8) continue;
            off.push_back(bo);
            cnt.push_back(e.count);
        }
        if (!off.empty()) (void)ns_segment_build_skips(ctx, dev, off.data(), cnt.data(), (uint32_t)off.size());
    }
    *out = dev;
    return true;
}

// reload (src/api_engine.cpp:50-160).  Like the reference, the new state replaces the old one only after EVERYTHING
// loaded (:76-90): segments, metadata and embeddings are built aside, the device copy goes into a FRESH context, and
// only then are the old context and the old host state dropped.  A failed reload leaves the engine serving what it
// served before.  (Old and new device copies coexist during the swap: 2 x index size of HBM, against 288 GB.)
bool Engine::reload() {
    std::lock_guard<std::recursive_mutex> lock(mtx_);
    err_.clear();
    // manifest, else scan segments/seg_* sorted (src/api_engine.cpp:57-73)
    std::vector<std::string> names = nsx::load_manifest(index_dir / "manifest.bin");
    if (names.empty()) {
        nsx::fs::path segroot = index_dir / "segments";
        std::error_code ec;
        if (nsx::fs::is_directory(segroot, ec)) {
            for (auto it = nsx::fs::directory_iterator(segroot, ec); !ec && it != nsx::fs::directory_iterator(); it.increment(ec)) {
                if (!it->is_directory(ec)) continue;
                auto name = it->path().filename().string();
                if (name.rfind("seg_", 0) == 0) names.push_back(name);
            }
            std::sort(names.begin(), names.end());
        }
    }
    if (names.empty()) { err_ = "no segments found under " + index_dir.string(); return false; }

    std::vector<nsx::SegmentData> loaded(names.size());
    for (size_t i = 0; i < names.size(); i++) {
        nsx::fs::path segdir = index_dir / "segments" / names[i];
        if (!nsx::load_segment(segdir, loaded[i])) { err_ = "failed to load segment: " + segdir.string(); return false; }
    }
    // metadata mapping (src/api_engine.cpp:110-113): absent file = no decoration, not an error
    nsx::MetadataTable fresh_meta;
    fresh_meta.load(index_dir / "metadata.csv", loaded);
    // embeddings (src/api_engine.cpp:115-153): only the terms some lexicon holds; absent or unusable file = no expansion
    nsx::SemanticTable fresh_sem;
    {
        nsx::fs::path emb;
        if (const char* p = std::getenv("EMBEDDINGS_PATH")) {
            emb = nsx::fs::path(p);
        } else {
            for (const char* c : {"embeddings.vec", "embeddings.txt", "glove.txt", "vectors.txt"})
                if (nsx::fs::exists(index_dir / c)) { emb = index_dir / c; break; }
        }
        if (!emb.empty() && nsx::fs::exists(emb)) {
            std::unordered_set<std::string> needed;
            needed.reserve(250000);
            for (const auto& seg : loaded)
                for (const auto& kv : seg.lex) needed.insert(kv.first);
            fresh_sem.load_from_text(emb, needed);
        }
    }

    ns_ctx* fresh_ctx = nullptr;
    std::vector<ns_seg*> fresh_segs;
    std::vector<Replica> fresh_replicas;
    if (device_ >= 0) {
        int rc = ns_ctx_create(device_, &fresh_ctx);
        if (rc != NS_OK) { err_ = std::string("ns_ctx_create: ") + ns_last_error(nullptr); return false; }
        // upload hook: once per segment, after load_segment (src/api_engine.cpp:90)
        fresh_segs.assign(loaded.size(), nullptr);
        bool ok = true;
        for (size_t i = 0; i < loaded.size() && ok; i++) ok = upload_segment(fresh_ctx, (uint32_t)i, loaded[i], &fresh_segs[i], err_);
        if (ok && fresh_sem.enabled) {
            rc = ns_sem_upload(fresh_ctx, fresh_sem.vecs.data(), (uint32_t)fresh_sem.terms.size(), (uint32_t)fresh_sem.dim, &fresh_sem.dev);
            if (rc != NS_OK) { err_ = std::string("ns_sem_upload: ") + ns_last_error(fresh_ctx); ok = false; }
        }
        // the replicas of a multi-device engine: the same upload into a fresh context on each further device
        for (size_t r = 0; r < replicas_.size() && ok; r++) {
            Replica fr;
            fr.device = replicas_[r].device;
            rc = ns_ctx_create(fr.device, &fr.ctx);
            if (rc != NS_OK) { err_ = std::string("ns_ctx_create (device ") + std::to_string(fr.device) + "): " + ns_last_error(nullptr); ok = false; break; }
            fr.segs.assign(loaded.size(), nullptr);
            for (size_t i = 0; i < loaded.size() && ok; i++) ok = upload_segment(fr.ctx, (uint32_t)i, loaded[i], &fr.segs[i], err_);
            fresh_replicas.push_back(fr);
        }
        if (!ok) {   // the engine keeps serving the previous index
            ns_ctx_destroy(fresh_ctx);   // frees the segments it holds
            for (auto& fr : fresh_replicas) if (fr.ctx) ns_ctx_destroy(fr.ctx);
            return false;
        }
    }
    // ---- commit ----
    release_device_segments();
    if (ctx_) ns_ctx_destroy(ctx_);
    ctx_ = fresh_ctx;
    dev_segs_ = std::move(fresh_segs);
    if (device_ >= 0) {
        for (auto& r : replicas_) if (r.ctx) ns_ctx_destroy(r.ctx);
        replicas_ = std::move(fresh_replicas);
    }
    seg_names = std::move(names);
    segments = std::move(loaded);
    meta = std::move(fresh_meta);
    sem = std::move(fresh_sem);
    dict.build(segments, [](uint32_t N, uint32_t df) { return bm25_idf(N, df); });
    cache_.clear();
    lru_.clear();
    raw_postings_.clear();
    // warm-up: the first requests after a reload otherwise pay for loading the kernels' code objects and for the
    // pinned staging buffers (~60 ms spread over the first few hundred requests).  One lone query over the longest
    // list of the first segment takes the same route (k_pull, k_uscore, k_merge_wide) once, here.
    const char* wu = std::getenv("NS_RELOAD_WARMUP");   // "0": skip (bench.py does, so that a profile of it holds the timed launches only)
    if (ctx_ && !segments.empty() && !(wu && wu[0] == '0')) {
        const nsx::LexEntry* best = nullptr;
        for (const auto& kv : segments[0].lex)
            if (kv.second.df && (!best || kv.second.count > best->count)) best = &kv.second;
        if (best) {
            ns_term_ref r;
            r.seg_id = 0; r.count = best->count; r.byte_off = segments[0].list_byte_offset(*best);
            r.idf = bm25_idf(segments[0].N, best->df); r.qweight = 1.0f;
            ns_query_desc qd{0, 1};
            ns_hit hits[10]; uint32_t nh = 0; uint64_t fd = 0;
            (void)ns_search_batch(ctx_, &qd, &r, 1, 10, hits, &nh, &fd, NS_FLAG_OR);
        }
    }
    return true;
}

const std::vector<uint8_t>* Engine::raw_postings(uint32_t seg) {
    std::lock_guard<std::recursive_mutex> lock(mtx_);
    if (seg >= segments.size()) return nullptr;
    if (raw_postings_.size() < segments.size()) raw_postings_.resize(segments.size());
    auto& slot = raw_postings_[seg];
    if (!slot) {
        slot = std::make_unique<std::vector<uint8_t>>();
        if (!nsx::read_postings(segments[seg], *slot)) { slot.reset(); err_ = "cannot read the inverted files of " + segments[seg].dir.string(); return nullptr; }
    }
    return slot.get();
}

bool Engine::build_impacts() {
    std::lock_guard<std::recursive_mutex> lock(mtx_);
    if (!ctx_) { err_ = "no device context"; return false; }
    for (size_t sid = 0; sid < segments.size(); sid++) {
        const auto& seg = segments[sid];
        std::vector<uint64_t> off;
        std::vector<uint32_t> cnt;
        std::vector<float> idf;
        off.reserve(seg.lex.size()); cnt.reserve(seg.lex.size()); idf.reserve(seg.lex.size());
        for (const auto& kv : seg.lex) {
            const nsx::LexEntry& e = kv.second;
            if (e.df == 0 || e.count == 0) continue;   // never scored (src/api_engine.cpp:458)
            off.push_back(seg.list_byte_offset(e));
            cnt.push_back(e.count);
            idf.push_back(bm25_idf(seg.N, e.df));   // the idf build_refs_range hands to the device for this list
        }
        int rc = ns_segment_build_impacts(ctx_, dev_segs_[sid], off.data(), cnt.data(), idf.data(), (uint32_t)off.size());
        if (rc != NS_OK) { err_ = std::string("ns_segment_build_impacts: ") + ns_last_error(ctx_); return false; }
        for (auto& r : replicas_) {
            rc = ns_segment_build_impacts(r.ctx, r.segs[sid], off.data(), cnt.data(), idf.data(), (uint32_t)off.size());
            if (rc != NS_OK) { err_ = std::string("ns_segment_build_impacts: ") + ns_last_error(r.ctx); return false; }
        }
    }
    return true;
}

// Optional: the compressed, blocked posting stream of every segment (ns_segment_build_packed).
bool Engine::build_packed() {
    std::lock_guard<std::recursive_mutex> lock(mtx_);
    if (!ctx_) { err_ = "no device context"; return false; }
    for (size_t sid = 0; sid < dev_segs_.size(); sid++) {
        int rc = ns_segment_build_packed(ctx_, dev_segs_[sid]);
        if (rc != NS_OK) { err_ = std::string("ns_segment_build_packed: ") + ns_last_error(ctx_); return false; }
        for (auto& r : replicas_) {
            rc = ns_segment_build_packed(r.ctx, r.segs[sid]);
            if (rc != NS_OK) { err_ = std::string("ns_segment_build_packed: ") + ns_last_error(r.ctx); return false; }
        }
    }
    return true;
}

// Optional: block maxima of every list long enough for pruning to pay (ns_segment_build_blockmax).
bool Engine::build_blockmax() {
    std::lock_guard<std::recursive_mutex> lock(mtx_);
    if (!ctx_) { err_ = "no device context"; return false; }
    for (size_t sid = 0; sid < segments.size(); sid++) {
        const auto& seg = segments[sid];
        std::vector<uint64_t> off;
        std::vector<uint32_t> cnt;
        std::vector<float> idf;
        for (const auto& kv : seg.lex) {
            const nsx::LexEntry& e = kv.second;
            if (e.df == 0 || e.count < 512) continue;
            if (seg.use_barrels && e.barrelId >= seg.barrel_base.size()) continue;
            const uint64_t bo = seg.list_byte_offset(e);
            if (bo % 8 != 0 || bo / 8 + e.count > seg.postings_bytes / 8) continue;   // a damaged record: never scored either
            off.push_back(bo);
            cnt.push_back(e.count);
            idf.push_back(bm25_idf(seg.N, e.df));   // the idf build_refs hands to the device for this list
        }
        int rc = ns_segment_build_blockmax(ctx_, dev_segs_[sid], off.data(), cnt.data(), idf.data(), (uint32_t)off.size());
        if (rc != NS_OK) { err_ = std::string("ns_segment_build_blockmax: ") + ns_last_error(ctx_); return false; }
        for (auto& r : replicas_) {
            rc = ns_segment_build_blockmax(r.ctx, r.segs[sid], off.data(), cnt.data(), idf.data(), (uint32_t)off.size());
            if (rc != NS_OK) { err_ = std::string("ns_segment_build_blockmax: ") + ns_last_error(r.ctx); return false; }
        }
    }
    return true;
}

void Engine::use_pruning(bool on) {
    std::lock_guard<std::recursive_mutex> lock(mtx_);
    if (ctx_) ns_ctx_use_pruning(ctx_, on ? 1 : 0);
    for (auto& r : replicas_) ns_ctx_use_pruning(r.ctx, on ? 1 : 0);
}

void Engine::use_merge(bool on) {
    std::lock_guard<std::recursive_mutex> lock(mtx_);
    if (ctx_) ns_ctx_use_merge(ctx_, on ? 1 : 0);
    for (auto& r : replicas_) ns_ctx_use_merge(r.ctx, on ? 1 : 0);
}

void Engine::share_scores(int mode) {
    std::lock_guard<std::recursive_mutex> lock(mtx_);
    if (ctx_) ns_ctx_share_scores(ctx_, mode);
    for (auto& r : replicas_) ns_ctx_share_scores(r.ctx, mode);
}

void Engine::use_packed(int mode) {
    std::lock_guard<std::recursive_mutex> lock(mtx_);
    if (ctx_) ns_ctx_use_packed(ctx_, mode);
    for (auto& r : replicas_) ns_ctx_use_packed(r.ctx, mode);
}

void Engine::use_skips(bool on) {
    std::lock_guard<std::recursive_mutex> lock(mtx_);
    if (ctx_) ns_ctx_use_skips(ctx_, on ? 1 : 0);
    for (auto& r : replicas_) ns_ctx_use_skips(r.ctx, on ? 1 : 0);
}

void Engine::use_impacts(bool on) {
    std::lock_guard<std::recursive_mutex> lock(mtx_);
    if (ctx_) ns_ctx_use_impacts(ctx_, on ? 1 : 0);
    for (auto& r : replicas_) ns_ctx_use_impacts(r.ctx, on ? 1 : 0);
}

// Query preparation (tokenise, stop-words, lexicon probes, idf: src/api_engine.cpp:388-397,:454-461) for
// queries [q0, q1) with GIVEN weighted terms (semantic expansion): `refs` receives the term refs of those queries,
// qd[q].term_begin is relative to it.  Terms are probed in the dictionary, once per term for all segments.
void Engine::build_refs_range(const std::vector<std::string>& queries, size_t q0, size_t q1, std::vector<ns_query_desc>& qd,
                              std::vector<ns_term_ref>& refs, std::vector<uint8_t>& usable,
                              const std::vector<nsx::WeightedTerms>* expanded) const {
    std::vector<int64_t> gid;
    for (size_t q = q0; q < q1; q++) {
        nsx::WeightedTerms own;
        if (!expanded)   // weights: 1.0f per base term (src/api_engine.cpp:419-421)
            for (auto& t : base_terms(queries[q])) own.emplace_back(std::move(t), 1.0f);
        const nsx::WeightedTerms& terms = expanded ? (*expanded)[q] : own;
        qd[q].term_begin = (uint32_t)refs.size();
        if (terms.empty() || segments.empty()) continue;   // src/api_engine.cpp:407, :424
        usable[q] = 1;
        gid.clear();
        for (const auto& tw : terms) gid.push_back(dict.find(tw.first.data(), tw.first.size()));
        for (uint32_t sid = 0; sid < segments.size(); sid++) {
            for (size_t ti = 0; ti < terms.size(); ti++) {
                if (gid[ti] < 0) continue;                    // :455
                const nsx::TermSeg& e = dict.row((uint32_t)gid[ti])[sid];
                if (e.byte_off == nsx::kAbsent) continue;     // :455 / :458
                ns_term_ref r;
                r.seg_id = sid;
                r.count = e.count;
                r.byte_off = e.byte_off;
                r.idf = e.idf;
                r.qweight = terms[ti].second;
                refs.push_back(r);
            }
        }
        qd[q].term_count = (uint32_t)refs.size() - qd[q].term_begin;
    }
}

// The same for plain base terms (weight 1.0f, src/api_engine.cpp:419-421) straight from the query bytes: no std::string
// per token, one dictionary probe per term, the per-segment numbers read from the term's row.
void Engine::build_refs_views(const QueryView* queries, size_t q0, size_t q1, ns_query_desc* qd, std::vector<ns_term_ref>& refs,
                              uint8_t* usable, std::vector<char>& scratch, std::vector<uint32_t>& gids) const {
    const uint32_t S = (uint32_t)segments.size();
    for (size_t q = q0; q < q1; q++) {
        ns_query_desc& d = qd[q - q0];
        d.term_begin = (uint32_t)refs.size();
        d.term_count = 0;
        gids.clear();
        size_t n_terms = 0;
        nsx::for_each_base_term(queries[q].p, queries[q].n, scratch, [&](const char* p, size_t n) {
            n_terms++;
            const int64_t g = dict.find(p, n);
            if (g >= 0) gids.push_back((uint32_t)g);
        });
        usable[q - q0] = (n_terms != 0 && S != 0) ? 1 : 0;   // src/api_engine.cpp:407: no base terms (or no segments) -> early return
        if (!usable[q - q0]) continue;
        for (uint32_t sid = 0; sid < S; sid++) {
            for (const uint32_t g : gids) {
                const nsx::TermSeg& e = dict.row(g)[sid];
                if (e.byte_off == nsx::kAbsent) continue;
                refs.push_back(ns_term_ref{sid, e.count, e.byte_off, e.idf, 1.0f});
            }
        }
        d.term_count = (uint32_t)refs.size() - d.term_begin;
    }
}

unsigned Engine::prep_width(size_t Q) const {
    unsigned nt = std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 16u);
    return (unsigned)std::max<size_t>(1, std::min<size_t>(nt, Q / 256));   // below ~256 queries per thread the hand-over costs more than it saves
}

// queries [q0, q1) on the engine's host threads (contiguous slices; the dictionary is read-only), the slices' term refs
// concatenated in query order.  qd / usable are indexed from q0.
void Engine::build_refs_parallel(const QueryView* queries, size_t q0, size_t q1, std::vector<ns_query_desc>& qd,
                                 std::vector<ns_term_ref>& refs, uint8_t* usable) const {
    const size_t Q = q1 - q0;
    qd.resize(Q);
    refs.clear();
    const unsigned nt = prep_width(Q);
    if (scratch_.size() < nt) scratch_.resize(nt);
    if (nt <= 1) {
        build_refs_views(queries, q0, q1, qd.data(), refs, usable, scratch_[0].text, scratch_[0].gids);
        return;
    }
    if (!pool_ || pool_->width() < nt) pool_.reset(new ForkJoin(std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 16u)));
    pool_->run(nt, [&](unsigned i) {
        PrepScratch& sc = scratch_[i];
        sc.refs.clear();
        const size_t a = q0 + Q * i / nt, b = q0 + Q * (i + 1) / nt;
        build_refs_views(queries, a, b, qd.data() + (a - q0), sc.refs, usable + (a - q0), sc.text, sc.gids);
    });
    size_t total = 0;
    std::vector<size_t> base(nt);
    for (unsigned i = 0; i < nt; i++) { base[i] = total; total += scratch_[i].refs.size(); }
    refs.resize(total);
    pool_->run(nt, [&](unsigned i) {
        const size_t a = Q * i / nt, b = Q * (i + 1) / nt;
        for (size_t q = a; q < b; q++) qd[q].term_begin += (uint32_t)base[i];
        if (!scratch_[i].refs.empty()) std::memcpy(refs.data() + base[i], scratch_[i].refs.data(), scratch_[i].refs.size() * sizeof(ns_term_ref));
    });
}

// A batch is Q independent searches: large batches are prepared by several host
// threads, each on a contiguous slice of the queries (the lexicons are read-only), and the slices'
// term refs are concatenated in query order.
void Engine::build_refs(const std::vector<std::string>& queries, std::vector<ns_query_desc>& qd,
                        std::vector<ns_term_ref>& refs, std::vector<uint8_t>& usable) const {
    std::lock_guard<std::recursive_mutex> lock(mtx_);
    const size_t Q = queries.size();
    qd.assign(Q, ns_query_desc{0, 0});
    usable.assign(Q, 0);
    refs.clear();
    refs_failed_ = false;
    if (sem.enabled) {   // src/api_engine.cpp:409-417: one device call per top-k size for the whole batch
        std::vector<nsx::WeightedTerms> expanded;
        if (!expand_queries(queries, expanded)) { refs_failed_ = true; return; }
        build_refs_range(queries, 0, Q, qd, refs, usable, &expanded);
        return;
    }
    std::vector<QueryView> views(Q);
    for (size_t q = 0; q < Q; q++) views[q] = QueryView{queries[q].data(), queries[q].size()};
    build_refs_parallel(views.data(), 0, Q, qd, refs, usable.data());
}

bool Engine::expand_queries(const std::vector<std::string>& queries, std::vector<nsx::WeightedTerms>& out) const {
    std::lock_guard<std::recursive_mutex> lock(mtx_);
    std::vector<std::vector<std::string>> qt(queries.size());
    for (size_t q = 0; q < queries.size(); q++) qt[q] = base_terms(queries[q]);
    if (!sem.enabled) {
        out.assign(queries.size(), {});
        for (size_t q = 0; q < queries.size(); q++)
            for (auto& t : qt[q]) out[q].emplace_back(t, 1.0f);
        return true;
    }
    if (!ctx_ || !sem.dev) { err_ = "embeddings are loaded but there is no device context: the similarity search has no CPU path"; return false; }
    return sem.expand_batch(ctx_, qt, out, err_);
}

bool Engine::prepare(const std::vector<std::string>& queries, int k, uint32_t flags, ns_batch** out) {
    std::lock_guard<std::recursive_mutex> lock(mtx_);
    if (!ctx_) { err_ = "no device context: this engine has no CPU scoring path"; return false; }
    const int K = std::max(1, std::min(k, 100));   // src/api_engine.cpp:377
    std::vector<ns_query_desc> qd;
    std::vector<ns_term_ref> refs;
    std::vector<uint8_t> usable;
    build_refs(queries, qd, refs, usable);
    if (refs_failed_) return false;
    int rc = ns_batch_prepare(ctx_, qd.data(), refs.data(), (uint32_t)queries.size(), (uint32_t)K, flags, out);
    if (rc != NS_OK) { err_ = std::string("ns_batch_prepare: ") + ns_last_error(ctx_); return false; }
    return true;
}

bool Engine::search_batch(const std::vector<std::string>& queries, int k, uint32_t flags, std::vector<SearchResult>& out) {
    std::lock_guard<std::recursive_mutex> lock(mtx_);
    return search_batch_locked(queries, k, flags, out);
}

// Sub-batches of a large batch (search_batch_flat): big enough to keep the device near its full-batch rate, small enough
// that preparing the first one — the only host work the device does not hide — is a small part of the call.  Measured on
// cfg5's 16384 queries (profiles/r03, ns_tool facade-bench): sub-batches of 2048 / 4096 / 5462 / 8192 / no cut =
// 8.4 / 5.0 / 4.6 / 3.65 / 3.83 ms per call.
static constexpr size_t kSubBatchDefault = 8192;
static size_t sub_batch_size() {   // NS_SUBBATCH overrides it (experiments)
    static const size_t v = []() {
        const char* e = std::getenv("NS_SUBBATCH");
        const long n = e ? std::atol(e) : 0;
        return n >= 64 ? (size_t)n : kSubBatchDefault;
    }();
    return v;
}

// One contiguous range [q0, q1) of a batch on one context.  Two or more sub-batches: prepare(i + 1) on the host || kernels(i)
// on the device || results(i - 1) on their way back.  pooled_prep: query preparation on the engine's host threads (the one
// range of a single-device engine); otherwise on the calling thread (a multi-device engine runs one such call per device).
bool Engine::run_range(ns_ctx* ctx, const QueryView* queries, size_t q0, size_t q1, int K, uint32_t flags, ns_hit* hits, uint32_t* nhits,
                       uint64_t* found, uint8_t* usable, bool pooled_prep, std::string& err) {
    const size_t Q = q1 - q0;
    if (Q == 0) return true;
    const size_t kSubBatch = sub_batch_size();
    const size_t n_sub = Q >= 2 * kSubBatch ? (Q + kSubBatch - 1) / kSubBatch : 1;
    const bool piped = n_sub > 1;
    if (piped) (void)ns_ctx_set_overlap(ctx, 1);
    struct InFlight { ns_batch* b = nullptr; size_t q0 = 0; };
    InFlight prev;
    bool ok = true;
    auto retire = [&](InFlight& f) {
        if (!f.b) return;
        if (ok) {
            const int rc = ns_batch_fetch(f.b, hits + f.q0 * (size_t)K, nhits + f.q0, found + f.q0);
            if (rc != NS_OK) { err = std::string("ns_batch_fetch: ") + ns_last_error(ctx); ok = false; }
        }
        ns_batch_destroy(f.b);
        f.b = nullptr;
    };
    std::vector<ns_query_desc> own_qd;
    std::vector<ns_term_ref> own_refs;
    PrepScratch own_sc;
    std::vector<ns_query_desc>& qd = pooled_prep ? flat_qd_ : own_qd;
    std::vector<ns_term_ref>& refs = pooled_prep ? flat_refs_ : own_refs;
    for (size_t i = 0; i < n_sub && ok; i++) {
        const size_t a = q0 + Q * i / n_sub, b = q0 + Q * (i + 1) / n_sub;
        if (pooled_prep) build_refs_parallel(queries, a, b, qd, refs, usable + a);
        else {
            qd.resize(b - a);
            refs.clear();
            build_refs_views(queries, a, b, qd.data(), refs, usable + a, own_sc.text, own_sc.gids);
        }
        InFlight cur;
        cur.q0 = a;
        int rc = ns_batch_prepare(ctx, qd.data(), refs.data(), (uint32_t)(b - a), (uint32_t)K, flags, &cur.b);
        if (rc == NS_OK) rc = ns_batch_run(cur.b, NS_RUN_FETCH);
        if (rc != NS_OK) {
            err = std::string("ns_batch_prepare/run: ") + ns_last_error(ctx);
            ok = false;
            if (cur.b) ns_batch_destroy(cur.b);
            break;
        }
        retire(prev);     // waits for sub-batch i - 1 only; sub-batch i is already queued behind it
        prev = cur;
    }
    retire(prev);
    if (piped) (void)ns_ctx_set_overlap(ctx, 0);
    return ok;
}

bool Engine::search_batch_flat(const QueryView* queries, size_t Q, int k, uint32_t flags, ns_hit* hits, uint32_t* nhits,
                               uint64_t* found, uint8_t* usable) {
    std::lock_guard<std::recursive_mutex> lock(mtx_);
    if (!ctx_) { err_ = "no device context: this engine has no CPU scoring path"; return false; }
    if (Q && (!queries || !hits || !nhits || !found || !usable)) { err_ = "search_batch_flat: null argument"; return false; }
    const int K = std::max(1, std::min(k, 100));   // src/api_engine.cpp:377
    if (Q == 0) return true;
    const size_t n_dev = num_devices();
    if (sem.enabled) {
        // semantic expansion runs the whole batch through two device calls on the primary context first
        // (src/api_engine.cpp:409-417); the scoring of the expanded terms is then sharded like any other batch
        std::vector<std::string> qs(Q);
        for (size_t q = 0; q < Q; q++) qs[q].assign(queries[q].p, queries[q].n);
        std::vector<ns_query_desc> qd;
        std::vector<ns_term_ref> refs;
        std::vector<uint8_t> us;
        build_refs(qs, qd, refs, us);
        if (refs_failed_) return false;
        std::memcpy(usable, us.data(), Q);
        std::vector<std::string> errs(n_dev);
        std::vector<int> rcs(n_dev, NS_OK);
        auto shard = [&](size_t d) {
            const auto [a, b] = n_dev > 1 ? shard_bounds(Q, d, n_dev) : std::pair<size_t, size_t>{0, Q};
            if (a >= b) return;
            ns_ctx* c = d == 0 ? ctx_ : replicas_[d - 1].ctx;
            // the shard's descriptors: term_begin stays an index into the one refs array
            rcs[d] = ns_search_batch(c, qd.data() + a, refs.data(), (uint32_t)(b - a), (uint32_t)K, hits + a * (size_t)K, nhits + a, found + a, flags);
            if (rcs[d] != NS_OK) errs[d] = std::string("ns_search_batch: ") + ns_last_error(c);
        };
        if (n_dev > 1 && Q >= 2 * n_dev) {
            std::vector<std::thread> th;
            for (size_t d = 1; d < n_dev; d++) th.emplace_back(shard, d);
            shard(0);
            for (auto& t : th) t.join();
        } else {
            const size_t keep = n_dev; (void)keep;
            rcs[0] = ns_search_batch(ctx_, qd.data(), refs.data(), (uint32_t)Q, (uint32_t)K, hits, nhits, found, flags);
            if (rcs[0] != NS_OK) errs[0] = std::string("ns_search_batch: ") + ns_last_error(ctx_);
        }
        for (size_t d = 0; d < n_dev; d++) if (rcs[d] != NS_OK) { err_ = errs[d]; return false; }
        return true;
    }
    if (n_dev > 1 && Q >= 2 * n_dev) {
        // one host thread + context per device, contiguous shards of ceil(Q / N) queries
        std::vector<std::string> errs(n_dev);
        std::vector<char> oks(n_dev, 1);
        auto shard = [&](size_t d) {
            const auto [a, b] = shard_bounds(Q, d, n_dev);
            ns_ctx* c = d == 0 ? ctx_ : replicas_[d - 1].ctx;
            oks[d] = run_range(c, queries, a, b, K, flags, hits, nhits, found, usable, /*pooled_prep*/ false, errs[d]) ? 1 : 0;
        };
        std::vector<std::thread> th;
        for (size_t d = 1; d < n_dev; d++) th.emplace_back(shard, d);
        shard(0);
        for (auto& t : th) t.join();
        for (size_t d = 0; d < n_dev; d++) if (!oks[d]) { err_ = errs[d]; return false; }
        return true;
    }
    return run_range(ctx_, queries, 0, Q, K, flags, hits, nhits, found, usable, /*pooled_prep*/ true, err_);
}

bool Engine::search_batch_locked(const std::vector<std::string>& queries, int k, uint32_t flags, std::vector<SearchResult>& out) {
    out.clear();
    if (!ctx_) { err_ = "no device context: this engine has no CPU scoring path"; return false; }
    const int K = std::max(1, std::min(k, 100));
    const size_t Q = queries.size();
    std::vector<QueryView> views(Q);
    for (size_t q = 0; q < Q; q++) views[q] = QueryView{queries[q].data(), queries[q].size()};
    std::vector<ns_hit> hits(Q * (size_t)K);
    std::vector<uint32_t> nhits(Q);
    std::vector<uint64_t> found(Q);
    std::vector<uint8_t> usable(Q);
    if (!search_batch_flat(views.data(), Q, k, flags, hits.data(), nhits.data(), found.data(), usable.data())) return false;
    out.resize(Q);
    for (size_t q = 0; q < Q; q++) {
        SearchResult& r = out[q];
        r.query = queries[q];
        r.k = K;
        r.segments = (int)segments.size();
        r.has_found = usable[q] != 0;
        if (!r.has_found) continue;
        r.found = found[q];
        r.hits.reserve(nhits[q]);
        for (uint32_t i = 0; i < nhits[q]; i++) {
            const ns_hit& h = hits[q * (size_t)K + i];
            r.hits.push_back(SearchHit{h.score, h.seg_id, h.doc_id});
        }
    }
    return true;
}

bool Engine::search_hits(const std::string& query, int k, uint32_t flags, SearchResult& out) {
    std::lock_guard<std::recursive_mutex> lock(mtx_);
    return search_hits_locked(query, k, flags, out);
}

bool Engine::search_hits_locked(const std::string& query, int k, uint32_t flags, SearchResult& out) {
    std::vector<SearchResult> v;
    if (!search_batch_locked({query}, k, flags, v)) return false;
    out = std::move(v[0]);
    return true;
}

std::string Engine::to_json(const SearchResult& r) const {
    std::lock_guard<std::recursive_mutex> lock(mtx_);
    return to_json_impl(r);
}

// no lock: search_batch_json runs it on several threads while the calling thread holds the engine lock
std::string Engine::to_json_impl(const SearchResult& r) const {
    std::string o;
    o += "{\n";
    if (r.has_found) o += "  \"found\": " + std::to_string(r.found) + ",\n";
    o += "  \"k\": " + std::to_string(r.k) + ",\n";
    o += "  \"query\": ";
    json_escape(o, r.query);
    o += ",\n";
    if (r.hits.empty()) {
        o += "  \"results\": [],\n";
    } else {
        o += "  \"results\": [\n";
        for (size_t i = 0; i < r.hits.size(); i++) {
            const SearchHit& h = r.hits[i];
            o += "    {\n";
            // result decoration (src/api_engine.cpp:516-531): only non-empty fields; keys in nlohmann's (alphabetical) order
            const nsx::MetaFields* md = meta.get(h.seg, h.doc);
            if (md && !md->author.empty()) { o += "      \"author\": "; json_escape(o, md->author); o += ",\n"; }
            o += "      \"cord_uid\": ";
            const auto& seg = segments[h.seg];
            json_escape(o, h.doc < seg.cord_uid.size() ? seg.cord_uid[h.doc] : std::string());
            o += ",\n";
            o += "      \"docId\": " + std::to_string(h.doc) + ",\n";
            if (md && !md->publish_time.empty()) { o += "      \"publish_time\": "; json_escape(o, md->publish_time); o += ",\n"; }
            o += "      \"score\": ";
            json_number_from_float(o, h.score);
            o += ",\n";
            o += "      \"segment\": ";
            json_escape(o, seg_names[h.seg]);
            if (md && !md->title.empty()) { o += ",\n      \"title\": "; json_escape(o, md->title); }
            if (md && !md->url.empty()) { o += ",\n      \"url\": "; json_escape(o, md->url); }
            o += "\n";
            o += (i + 1 < r.hits.size()) ? "    },\n" : "    }\n";
        }
        o += "  ],\n";
    }
    o += "  \"segments\": " + std::to_string(r.segments) + "\n";
    o += "}";
    return o;
}

bool Engine::search_batch_json(const std::vector<std::string>& queries, int k, std::vector<std::string>& out) {
    std::vector<SearchResult> res;
    std::lock_guard<std::recursive_mutex> lock(mtx_);
    if (!search_batch_locked(queries, k, NS_FLAG_OR, res)) return false;
    // result assembly reads segments / meta, which only reload() replaces: the lock is held to the end
    const size_t Q = res.size();
    out.assign(Q, std::string());
    unsigned nt = std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 16u);
    nt = (unsigned)std::min<size_t>(nt, Q / 512);
    if (nt <= 1) {
        for (size_t q = 0; q < Q; q++) out[q] = to_json_impl(res[q]);
        return true;
    }
    std::vector<std::thread> th;
    for (unsigned i = 0; i < nt; i++)
        th.emplace_back([&, i]() { for (size_t q = Q * i / nt; q < Q * (i + 1) / nt; q++) out[q] = to_json_impl(res[q]); });
    for (auto& t : th) t.join();
    return true;
}

// Every engine entry takes the one engine lock, as the reference does (src/api_engine.cpp:372 `std::lock_guard<std::mutex>
// lock(mtx)`; :54, :168): the HTTP layer calls search() from a thread pool, and the result cache, the error string and
// the device context (one stream, one set of pinned staging buffers: include/nextsearch_hip.h) are not re-entrant.
bool Engine::search_text(const std::string& query, int k, std::string& body) {
    std::lock_guard<std::recursive_mutex> lock(mtx_);
    const int K = std::max(1, std::min(k, 100));
    const std::string key = query + "|" + std::to_string(K);            // make_cache_key (:190-192), K already clamped (:377-380)
    if (cache_on_) {
        auto it = cache_.find(key);
        if (it != cache_.end()) {                                       // get_from_cache (:195-210)
            lru_.erase(it->second.lru);
            lru_.push_front(key);
            it->second.lru = lru_.begin();
            // result["from_cache"] = true: nlohmann keeps keys sorted, so the flag sits right before "k"
            body = it->second.body;
            const size_t at = body.find("  \"k\": ");
            if (at != std::string::npos) body.insert(at, "  \"from_cache\": true,\n");
            return true;
        }
    }
    SearchResult r;
    if (!search_hits_locked(query, k, NS_FLAG_OR, r)) { body = err_; return false; }   // the message, for the caller that has no other way to it
    body = to_json_impl(r);
    if (cache_on_ && r.has_found) {                                     // put_in_cache (:213-250); the early returns (:407,:424) skip it
        if (cache_.size() >= kMaxCacheSize) {
            auto ev = cache_.find(lru_.back());
            if (ev != cache_.end()) { lru_.erase(ev->second.lru); cache_.erase(ev); }
        }
        lru_.push_front(key);
        cache_[key] = CacheEntry{body, lru_.begin()};
    }
    return true;
}

std::string Engine::search(const std::string& query, int k) {
    std::string body;
    if (!search_text(query, k, body)) {
        // the reference lets exceptions reach the HTTP layer's 500 handler (src/api_server.cpp:76-84)
        std::string o = "{\n  \"error\": ";
        json_escape(o, body);
        o += "\n}";
        return o;
    }
    return body;
}

}  // namespace nextsearch
