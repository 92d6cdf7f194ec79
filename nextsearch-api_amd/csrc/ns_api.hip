// C-ABI implementation of include/nextsearch_hip.h: context, segment upload (pinned staging),
// batch preparation (term groups -> work items), kernel launches and result fetch.
// No CPU fallback exists in this library: every compute entry point needs a live HIP device.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <numeric>
#include <string>
#include <thread>
#include <map>
#include <vector>

#include "../../include/nextsearch_hip.h"
#include "ns_internal.h"
#include "ns_forkjoin.hpp"
#include "ns_kernels.hip"
#include "ns_wave_kernel.hip"
#include "ns_driver_kernel.hip"
#include "ns_prune_kernel.hip"
#include "ns_merge_kernel.hip"
#include "ns_tile_kernel.hip"
#include "ns_invert.hip"
#include "ns_sem.hip"

using namespace ns;

static_assert(sizeof(ns_hit) == sizeof(Hit), "ns_hit layout");


struct ns_prep;
static void prep_free(ns_prep* p);

// ------------------------------------------------------------------------------------------------
struct ns_seg {
    ns_ctx* ctx = nullptr;
    uint32_t id = 0;
    uint32_t n_docs = 0;
    uint64_t n_postings = 0;
    uint2* d_postings = nullptr;
    float* d_norm = nullptr;    // per doc
    float* d_pnorm = nullptr;   // per posting
    bool norm_safe = false;     // every norm lies in [2^-20, 2^30]: the BM25 division may take its short form (ns_div_short)
    float avgdl = 0.0f;
    // upload in progress (ns_segment_upload_begin .. _end): payload bytes received so far, doc_len on the device, pinned staging
    bool pending = false;
    uint64_t filled = 0;
    uint32_t* d_len = nullptr;
    void* pin[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    int stage_k = 0;
    // packed posting stream (ns_segment_build_packed; ns_internal.h kPk*): blocks, per-block headers; the norm index per
    // doc and the table of distinct norms are made at upload (they only exist when the segment has <= 65536 distinct doc lengths)
    uint32_t* d_packed = nullptr;
    uint2* d_pk_hdr = nullptr;
    float* d_pk_scores = nullptr;   // per posting: the impact stream's score alone (built when both streams exist)
    uint16_t* d_nidx = nullptr;
    float* d_ntab = nullptr;
    uint32_t n_norms = 0;
    // Optional impact stream (ns_segment_build_impacts): {docId, term score bits} per posting of the registered lists,
    // index-aligned with d_postings.  `imp_tab`: open-addressed (first posting index -> count, idf bits) of those lists.
    uint2* d_impacts = nullptr;
    struct ImpList { uint32_t first = 0xFFFFFFFFu, count = 0, idf_bits = 0; };
    std::vector<ImpList> imp_tab;   // size is a power of two (or 0)
    size_t imp_lists = 0;
    // Optional skip tables (ns_segment_build_skips; DevSeg::skips): `skip_tab` maps a list's first posting index to its count and
    // to the index of its first table entry; every table has skip_entries() entries.
    uint32_t* d_skips = nullptr;
    std::vector<uint32_t*> skip_retired;    // outgrown blocks: batches prepared before the growth still point into them
    uint64_t skip_cap = 0, skip_used = 0;   // entries allocated / in use
    struct SkipList { uint32_t first = 0xFFFFFFFFu, count = 0, entry = 0; };
    std::vector<SkipList> skip_tab;   // open-addressed, size a power of two (or 0)
    size_t skip_lists = 0;
    uint32_t skip_entries() const { return (n_docs + kSkipDocs - 1) / kSkipDocs + 2; }
    // 1 + index of the first table entry of the list [first, first + count), or 0
    uint32_t skip_of(uint32_t first, uint32_t count) const {
        if (skip_tab.empty()) return 0;
        const size_t mask = skip_tab.size() - 1;
        for (size_t h = ((size_t)first * 0x9E3779B1u) & mask;; h = (h + 1) & mask) {
            const SkipList& e = skip_tab[h];
            if (e.first == first) return e.count == count ? e.entry + 1u : 0u;
            if (e.first == 0xFFFFFFFFu) return 0;
        }
    }
    // Optional block maxima (ns_segment_build_blockmax; DevSeg::blockmax): per registered list ceil(count / 256) fp32 values.
    float* d_blockmax = nullptr;
    std::vector<float*> bmx_retired;         // outgrown blocks: batches prepared before the growth still point into them
    uint64_t bmx_cap = 0, bmx_used = 0;      // entries allocated / in use
    struct BmxList { uint32_t first = 0xFFFFFFFFu, count = 0, idf_bits = 0, entry = 0; };
    std::vector<BmxList> bmx_tab;            // open-addressed, size a power of two (or 0)
    // 1 + index of the first block maximum of the list [first, first + count) built with this idf, or 0
    uint32_t bmx_of(uint32_t first, uint32_t count, uint32_t idf_bits) const {
        if (bmx_tab.empty()) return 0;
        const size_t mask = bmx_tab.size() - 1;
        for (size_t h = ((size_t)first * 0x9E3779B1u) & mask;; h = (h + 1) & mask) {
            const BmxList& e = bmx_tab[h];
            if (e.first == first) return (e.count == count && e.idf_bits == idf_bits) ? e.entry + 1u : 0u;
            if (e.first == 0xFFFFFFFFu) return 0;
        }
    }
    // Shared term scores (ns_ctx_share_scores): every list a sharing batch ever built into d_impacts, first -> count.  Lists
    // that overlap another one are refused (two builders would write the same postings with different values).
    std::map<uint32_t, uint32_t> share_lists;
    bool share_admit(uint32_t first, uint32_t count) {
        auto it = share_lists.lower_bound(first);
        if (it != share_lists.end() && (it->first == first ? it->second != count : (uint64_t)first + count > it->first)) return false;
        if (it != share_lists.begin() && (it == share_lists.end() || it->first != first)) {
            auto pv = std::prev(it);
            if ((uint64_t)pv->first + pv->second > first) return false;
        }
        share_lists.emplace(first, count);
        return true;
    }
    bool imp_has(uint32_t first, uint32_t count, uint32_t idf_bits) const {
        if (imp_tab.empty()) return false;
        const size_t mask = imp_tab.size() - 1;
        for (size_t h = ((size_t)first * 0x9E3779B1u) & mask;; h = (h + 1) & mask) {
            const ImpList& e = imp_tab[h];
            if (e.first == first) return e.count == count && e.idf_bits == idf_bits;
            if (e.first == 0xFFFFFFFFu) return false;
        }
    }
};

struct ns_ctx {
    // Device blocks of destroyed batches are kept for the next batch (a serving loop prepares batch after batch
    // of similar shape; 14 hipMalloc + 14 hipFree per batch cost more than the descriptors' upload).
    struct Block { void* p; size_t n; };
    std::vector<Block> pool;
    size_t pool_bytes = 0;
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    // Batches alternate between the ctx's stream and this second one when overlap is on (ns_ctx_set_overlap): the
    // head of batch i+1 then fills the wave slots that the draining tail of batch i leaves idle.
    hipStream_t alt_stream = nullptr;
    // Large descriptor uploads are pulled by a kernel on a stream of their own (highest priority), so that batch i+1's upload
    // runs NEXT TO batch i's scoring kernel instead of behind it on the batch's stream or in the shared DMA queue.
    hipStream_t pull_stream = nullptr;
    bool overlap = false, flip = false;
    std::string err;
    std::string devname;
    int n_cus = 0;
    std::vector<ns_seg*> segs;   // indexed by seg_id
    std::vector<ns_seg*> pending_uploads;   // begun (ns_segment_upload_begin), not yet ended or released: freed with the ctx
    uint32_t variant = 0;
    uint32_t min_items = 0;
    uint32_t split_postings = 0;
    // Pinned staging: a batch's descriptor arrays go up in ONE copy and its three result arrays come down in
    // ONE copy (a lone query is otherwise dominated by ten small pageable copies and the syncs they imply).
    void* h_up = nullptr;
    size_t h_up_cap = 0;
    hipEvent_t up_done = nullptr;   // recorded after the upload that reads h_up; waited on before h_up is rewritten
    bool up_busy = false;
    void* h_down = nullptr;
    size_t h_down_cap = 0;
    // A small batch has its result arrays IN h_down (pinned host memory is device-addressable): the kernels
    // write the few hits over PCIe themselves and fetch is a stream sync + memcpy.  One batch at a time owns it.
    struct ns_batch* down_owner = nullptr;
    // pinned result buffers for batches in flight (NS_RUN_FETCH): one per batch between its run and its fetch
    struct DownSlot { void* p = nullptr; size_t cap = 0; bool busy = false; };
    std::vector<DownSlot> down_slots;
    // Shared term scores (ns_ctx_share_scores): 0 off; 1 a batch whose term refs name each distinct list often enough computes
    // every list's BM25 term scores ONCE (k_share_scores, in front of the scoring kernel, on every run) and scores from
    // {docId, score}; 2 every batch that can (tests).  The registry maps (segment, first posting) to the list's count and idf and
    // to the last batch that listed it; `live_shared` counts the sharing batches alive: an idf may only change while it is 0.
    int share_mode = 1;
    uint32_t share_ratio = 48;           // share when postings >= share_ratio x distinct postings (below ~50 uses per posting the extra kernel costs what it saves) ...
    uint64_t share_min_postings = 4u << 20;   // ... and the batch scans at least this many postings
    struct ShareEnt { uint64_t key = ~0ull; uint32_t count = 0, idf_bits = 0, epoch = 0; bool bad = false; };
    std::vector<ShareEnt> share_tab;     // open-addressed, size a power of two (or 0)
    size_t share_n = 0;
    uint32_t share_epoch = 0;
    uint32_t live_shared = 0;
    bool use_impacts = true;   // batches take the impact stream when every list they touch has one (ns_ctx_use_impacts)
    int use_packed = 1;        // 0 off; 1, 2: batches read the packed stream when every segment they touch has one (ns_ctx_use_packed)
    bool use_skips = true;     // doc-tile groups walk the skip grid when their lists have skip tables (ns_ctx_use_skips)
    bool use_merge = true;     // general-class groups of exactly two term refs take the two-list merge body (ns_ctx_use_merge)
    uint32_t merge_ratio = 8;  // ... when the longer list is at most this many times the shorter (NS_MERGE_RATIO: sweeps)
    bool use_pruning = false;  // single-term groups whose list has block maxima skip the blocks that cannot enter the top-K (ns_ctx_use_pruning)
    ns_prep* prep = nullptr;   // ns_batch_prepare's host threads and per-thread scratch, kept from batch to batch
    unsigned prep_threads = 0; // 0 = automatic (up to 8); 1 = prepare on the calling thread only (ns_ctx_set_host_threads)
    // Launch order inside coarse run-time classes (see "XCD dealing" in ns_batch_prepare): 1 = on.  The environment variables
    // NS_ORDER_MODE (0 = off) / NS_ORDER_COARSE (log2 of the fine buckets per class) override it for experiments; read at
    // ns_ctx_create.
    int order_mode = 1, order_coarse = 3;
    uint32_t key_pct[4] = {100, 100, 100, 100};   // launch-order key of general / thin / tile / merge items in per cent (NS_KEY_PCT=g,t,d,m: sweeps)
    uint32_t tile_dens64 = 16;   // doc-tile class from this many postings per 64 docs (0.25 per doc); NS_TILE_DENS64 overrides (sweeps)
    bool order_coarse_forced = false;   // NS_ORDER_COARSE given: no automatic choice
};

static thread_local std::string g_create_err;
static void seg_free_device_fwd(ns_seg* s);
static void seg_free_staging_fwd(ns_seg* s);

static int fail(ns_ctx* ctx, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    std::vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf; else g_create_err = buf;
    return code;
}

#define HIPCHK(ctx, call)                                                                         \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return fail((ctx), NS_E_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// Kernel variants (DESIGN.md "kernel variants").
//   wave variants: hb = table entries per wave (k_dscore) or docs per tile (k_tscore)
//   workgroup variants (k_score, the >64-terms fallback and the round-1 baseline): threads per
//   workgroup, slots per thread, postings per thread per round; tile_docs = nt * spt.
struct VariantDesc { uint32_t hb; uint32_t nt, spt, u; uint32_t d; };
static const VariantDesc kVariants[] = {
    {512, 512, 12, 4, 0},       // 0: default = AUTO: k_uscore, per (query, segment) group the driver-stream body (64 or 192 foreign postings per super-batch) or 1024-doc tiles, by its mix of lists
    {0, 1024, 12, 4, 0},        // 1: workgroup kernel, 12288-doc tiles
    {0, 512, 12, 4, 0},         // 2: workgroup kernel,  6144-doc tiles
    {0, 256, 16, 4, 0},         // 3: workgroup kernel,  4096-doc tiles
    {0, 512, 16, 8, 0},         // 4: workgroup kernel,  8192-doc tiles
    {0, 0, 0, 0, 0},            // 5..11: retired in round 2 (the wave-private batch kernel k_wscore); ns_set_tuning rejects them
    {0, 0, 0, 0, 0}, {0, 0, 0, 0, 0}, {0, 0, 0, 0, 0}, {0, 0, 0, 0, 0}, {0, 0, 0, 0, 0}, {0, 0, 0, 0, 0},
    {512, 512, 12, 4, 0},       // 12: driver-stream kernel (k_dscore), 512 slots, 128 foreign postings per super-batch   [d == 0 marks k_dscore]
    {256, 512, 12, 4, 0},       // 13: k_dscore  256 slots /  64 foreign
    {1024, 512, 12, 4, 0},      // 14: k_dscore 1024 slots / 256 foreign
    {512, 512, 12, 4, 0},       // 15: k_dscore  512 slots /  64 foreign
    {512, 512, 12, 4, 0},       // 16: k_dscore  512 slots / 256 foreign
    {1024, 512, 12, 4, 0},      // 17: k_dscore 1024 slots / 128 foreign
    {512, 512, 12, 4, 1},       // 18: doc-tile kernel k_tscore for every group, 512-doc tiles   [d == 1 marks k_tscore]
    {1024, 512, 12, 4, 1},      // 19: k_tscore, 1024-doc tiles
    {2048, 512, 12, 4, 1},      // 20: k_tscore, 2048-doc tiles
};
static constexpr uint32_t kNumVariants = sizeof(kVariants) / sizeof(kVariants[0]);
static constexpr uint32_t kWaveMaxTerms = 64;
static constexpr uint32_t kDefaultSplitPostings = 32768;   // forced variants: postings per work item
// auto mode: work units per item (one unit = one streamed driver posting).  Every item pays for its own
// top-K warm-up and its K-row partial result, so large K wants fewer, longer items (sweeps: profiles/r01).
static constexpr uint32_t kSplitWorkSmallK = 98304, kSplitWorkLargeK = 131072;
static constexpr uint64_t kWorkForeign = 8, kWorkTile = 2, kWorkMerge = 4;   // merge: fitted on two-list laws (profiles/r03): 1.0 ps per unit, like the general class
static constexpr uint32_t kSkipMinCount = 64;   // shorter lists are never looked up in the skip registry (ns_segment_build_skips)
// per-item, per-term constants of the launch-order key (fitted to per-item timestamps, tools/dbg/item_times.py)
static constexpr uint64_t kItemTermGeneral = 4000, kItemTermThin = 3000, kItemTermTile = 8000;   // general re-fitted in round 2 (10000 -> 4000: ab16)

template <int HK, int FB>
static void launch_dscore(bool and_mode, uint32_t n_items, hipStream_t st, const DevWItem* items, const DevTerm* terms,
                          const DevSeg* segs, Hit* hits, uint32_t* nhits, uint64_t* found, uint32_t K) {
    dim3 grid((n_items + 3) / 4), block(256);
    if (and_mode)
        hipLaunchKernelGGL((k_dscore<HK, FB, true>), grid, block, 0, st, items, n_items, terms, segs, hits, nhits, found, K);
    else
        hipLaunchKernelGGL((k_dscore<HK, FB, false>), grid, block, 0, st, items, n_items, terms, segs, hits, nhits, found, K);
}

template <int TD>
static void launch_tscore(bool and_mode, uint32_t n_items, hipStream_t st, const DevWItem* items, const DevTerm* terms,
                          const DevSeg* segs, Hit* hits, uint32_t* nhits, uint64_t* found, uint32_t K) {
    dim3 grid((n_items + 3) / 4), block(256);
    if (and_mode)
        hipLaunchKernelGGL((k_tscore<TD, true>), grid, block, 0, st, items, n_items, terms, segs, hits, nhits, found, K);
    else
        hipLaunchKernelGGL((k_tscore<TD, false>), grid, block, 0, st, items, n_items, terms, segs, hits, nhits, found, K);
}

template <int NT, int SPT, int U>
static void launch_score(bool and_mode, uint32_t n_items, hipStream_t st, const DevItem* items, const DevTerm* terms,
                         const DevSeg* segs, const uint32_t* bounds, Hit* hits, uint32_t* nhits, uint64_t* found,
                         uint32_t K) {
    if (and_mode)
        hipLaunchKernelGGL((k_score<NT, SPT, U, true>), dim3(n_items), dim3(NT), 0, st, items, terms, segs, bounds, hits, nhits, found, K);
    else
        hipLaunchKernelGGL((k_score<NT, SPT, U, false>), dim3(n_items), dim3(NT), 0, st, items, terms, segs, bounds, hits, nhits, found, K);
}

// ------------------------------------------------------------------------------------------------
extern "C" int ns_ctx_create(int device, ns_ctx** out) {
    if (!out) return fail(nullptr, NS_E_INVAL, "ns_ctx_create: out is NULL");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, NS_E_NODEVICE, "no HIP device available (%s); this library has no CPU fallback",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (device < 0 || device >= ndev) return fail(nullptr, NS_E_INVAL, "device %d out of range [0,%d)", device, ndev);
    e = hipSetDevice(device);
    if (e != hipSuccess) return fail(nullptr, NS_E_NODEVICE, "hipSetDevice(%d): %s", device, hipGetErrorString(e));
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) return fail(nullptr, NS_E_NODEVICE, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    ns_ctx* ctx = new ns_ctx();
    ctx->device = device;
    ctx->devname = std::string(prop.gcnArchName) + " " + prop.name;
    ctx->n_cus = prop.multiProcessorCount;
    e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        int rc = fail(nullptr, NS_E_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
        delete ctx;
        return rc;
    }
    ctx->stream = ctx->own_stream;
    if (const char* om = std::getenv("NS_ORDER_MODE")) ctx->order_mode = std::atoi(om);
    if (const char* kp = std::getenv("NS_KEY_PCT")) {
        unsigned a = 100, b = 100, c = 100, d = 100;
        if (std::sscanf(kp, "%u,%u,%u,%u", &a, &b, &c, &d) >= 1) { ctx->key_pct[0] = a; ctx->key_pct[1] = b; ctx->key_pct[2] = c; ctx->key_pct[3] = d; }
    }
    if (const char* td = std::getenv("NS_TILE_DENS64")) ctx->tile_dens64 = (uint32_t)std::max(1, std::atoi(td));
    if (const char* um = std::getenv("NS_MERGE")) ctx->use_merge = std::atoi(um) != 0;
    if (const char* mr = std::getenv("NS_MERGE_RATIO")) ctx->merge_ratio = (uint32_t)std::max(1, std::atoi(mr));
    if (const char* sm = std::getenv("NS_SHARE")) ctx->share_mode = std::max(0, std::min(2, std::atoi(sm)));
    if (const char* sr = std::getenv("NS_SHARE_RATIO")) ctx->share_ratio = (uint32_t)std::max(1, std::atoi(sr));
    if (const char* sp = std::getenv("NS_SHARE_MIN")) ctx->share_min_postings = (uint64_t)std::max(0ll, std::atoll(sp));
    if (const char* oc = std::getenv("NS_ORDER_COARSE")) { ctx->order_coarse = std::max(0, std::min(11, std::atoi(oc))); ctx->order_coarse_forced = true; }
    if (hipStreamCreateWithFlags(&ctx->alt_stream, hipStreamNonBlocking) != hipSuccess) { ctx->alt_stream = nullptr; (void)hipGetLastError(); }
    {
        int lo_pri = 0, hi_pri = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo_pri, &hi_pri);
        if (hipStreamCreateWithPriority(&ctx->pull_stream, hipStreamNonBlocking, hi_pri) != hipSuccess) { ctx->pull_stream = nullptr; (void)hipGetLastError(); }
    }
    *out = ctx;
    return NS_OK;
}

extern "C" void ns_ctx_destroy(ns_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->alt_stream) (void)hipStreamSynchronize(ctx->alt_stream);
    if (ctx->pull_stream) (void)hipStreamSynchronize(ctx->pull_stream);
    for (ns_seg* s : ctx->segs) {
        if (!s) continue;
        seg_free_device_fwd(s);
        delete s;
    }
    for (ns_seg* s : ctx->pending_uploads) {   // uploads begun and never ended: their HBM, pinned staging and events go with the ctx
        seg_free_staging_fwd(s);
        seg_free_device_fwd(s);
        delete s;
    }
    for (auto& blk : ctx->pool) (void)hipFree(blk.p);
    if (ctx->h_up) (void)hipHostFree(ctx->h_up);
    if (ctx->h_down) (void)hipHostFree(ctx->h_down);
    for (auto& ds : ctx->down_slots) if (ds.p) (void)hipHostFree(ds.p);
    if (ctx->up_done) (void)hipEventDestroy(ctx->up_done);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    if (ctx->alt_stream) (void)hipStreamDestroy(ctx->alt_stream);
    if (ctx->pull_stream) (void)hipStreamDestroy(ctx->pull_stream);
    prep_free(ctx->prep);
    delete ctx;
}

extern "C" int ns_ctx_set_stream(ns_ctx* ctx, void* hip_stream) {
    if (!ctx) return fail(nullptr, NS_E_INVAL, "ns_ctx_set_stream: ctx is NULL");
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return NS_OK;
}

extern "C" const char* ns_last_error(ns_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }
extern "C" const char* ns_device_name(ns_ctx* ctx) { return ctx ? ctx->devname.c_str() : ""; }

extern "C" int ns_set_tuning(ns_ctx* ctx, uint32_t variant, uint32_t min_items, uint32_t split_postings) {
    if (!ctx) return fail(nullptr, NS_E_INVAL, "ns_set_tuning: ctx is NULL");
    if (variant >= kNumVariants || kVariants[variant].nt == 0) return fail(ctx, NS_E_INVAL, "unknown kernel variant %u", variant);
#ifndef NS_VARIANTS
    // The product library holds ONE scoring kernel (k_uscore, variant 0) plus k_score as the fallback for term groups of more
    // than 64 terms.  The forced variants — each body as a kernel of its own, other table / tile sizes — exist for the
    // parity tests and for sweeps: `make -C nextsearch-api_amd variants` builds libnextsearch_hip_variants.so with them.
    if (variant != 0) return fail(ctx, NS_E_INVAL, "kernel variant %u exists only in the variants build (make -C nextsearch-api_amd variants)", variant);
#endif
    ctx->variant = variant;
    ctx->min_items = min_items;
    ctx->split_postings = split_postings;
    return NS_OK;
}

// ------------------------------------------------------------------------------------------------
// slack behind the payload: the driver stream loads whole rounds of 256 postings (and their norms) from a list's
// cursor, i.e. up to 255 entries past the end of the last list
static constexpr size_t kPadPostings = 256;
static constexpr size_t kStageChunk = 32u << 20;   // pinned, double-buffered: the host memcpy of chunk i+1 overlaps the DMA of chunk i

static void seg_free_staging(ns_seg* s) {
    for (int i = 0; i < 2; i++) {
        if (s->pin[i]) (void)hipHostFree(s->pin[i]);
        if (s->ev[i]) (void)hipEventDestroy(s->ev[i]);
        s->pin[i] = nullptr; s->ev[i] = nullptr;
    }
    if (s->d_len) (void)hipFree(s->d_len);
    s->d_len = nullptr;
}
static void seg_free_device(ns_seg* s);
static void seg_free_device_fwd(ns_seg* s) { seg_free_device(s); }
static void seg_free_staging_fwd(ns_seg* s) { seg_free_staging(s); }
static void seg_free_device(ns_seg* s) {
    (void)hipFree(s->d_postings);
    (void)hipFree(s->d_norm);
    (void)hipFree(s->d_pnorm);
    (void)hipFree(s->d_impacts);
    (void)hipFree(s->d_packed);
    (void)hipFree(s->d_pk_hdr);
    (void)hipFree(s->d_pk_scores);
    (void)hipFree(s->d_nidx);
    (void)hipFree(s->d_ntab);
    (void)hipFree(s->d_skips);
    for (uint32_t* p : s->skip_retired) (void)hipFree(p);
    s->skip_retired.clear();
    (void)hipFree(s->d_blockmax);
    for (float* p : s->bmx_retired) (void)hipFree(p);
    s->bmx_retired.clear();
    s->d_blockmax = nullptr; s->bmx_cap = s->bmx_used = 0; s->bmx_tab.clear();
    s->d_skips = nullptr; s->skip_cap = s->skip_used = 0; s->skip_tab.clear(); s->skip_lists = 0;
    s->d_postings = nullptr; s->d_norm = nullptr; s->d_pnorm = nullptr; s->d_impacts = nullptr; s->d_packed = nullptr;
    s->d_pk_hdr = nullptr; s->d_pk_scores = nullptr; s->d_nidx = nullptr; s->d_ntab = nullptr;
}
// host bytes -> device through the segment's two pinned buffers
static hipError_t seg_stage(ns_ctx* ctx, ns_seg* s, void* dst, const void* src, size_t n) {
    size_t off = 0;
    while (off < n) {
        const size_t c = std::min(kStageChunk, n - off);
        const int k = s->stage_k;
        hipError_t r = hipEventSynchronize(s->ev[k]);
        if (r != hipSuccess) return r;
        std::memcpy(s->pin[k], (const char*)src + off, c);
        r = hipMemcpyAsync((char*)dst + off, s->pin[k], c, hipMemcpyHostToDevice, ctx->stream);
        if (r != hipSuccess) return r;
        r = hipEventRecord(s->ev[k], ctx->stream);
        if (r != hipSuccess) return r;
        off += c;
        s->stage_k ^= 1;
    }
    return hipSuccess;
}

extern "C" int ns_segment_upload_begin(ns_ctx* ctx, uint32_t seg_id, uint32_t n_docs, float avgdl, const uint32_t* doc_len,
                                       uint64_t nbytes, ns_seg** out) {
    if (!ctx) return fail(nullptr, NS_E_INVAL, "ns_segment_upload: ctx is NULL");
    if (!out) return fail(ctx, NS_E_INVAL, "ns_segment_upload_begin: out is NULL");
    *out = nullptr;
    if (seg_id >= (1u << 20)) return fail(ctx, NS_E_INVAL, "seg_id %u too large", seg_id);
    if (nbytes % 8 != 0) return fail(ctx, NS_E_INVAL, "posting payload of %llu bytes is not a whole number of {u32,u32} pairs", (unsigned long long)nbytes);
    if (nbytes / 8 >= (1ull << 32)) return fail(ctx, NS_E_INVAL, "segment has %llu postings; this build indexes postings with 32 bits (split the segment)", (unsigned long long)(nbytes / 8));
    if (n_docs && !doc_len) return fail(ctx, NS_E_INVAL, "null doc_len/postings");
    if (seg_id < ctx->segs.size() && ctx->segs[seg_id]) return fail(ctx, NS_E_INVAL, "segment %u already uploaded", seg_id);
    HIPCHK(ctx, hipSetDevice(ctx->device));

    ns_seg* s = new ns_seg();
    s->ctx = ctx;
    s->id = seg_id;
    s->n_docs = n_docs;
    s->n_postings = nbytes / 8;
    s->avgdl = avgdl;
    s->pending = true;
    auto cleanup = [&]() { seg_free_staging(s); seg_free_device(s); delete s; };

    hipError_t e;
    if ((e = hipMalloc((void**)&s->d_postings, nbytes + kPadPostings * 8)) != hipSuccess) { cleanup(); return fail(ctx, NS_E_NOMEM, "hipMalloc postings (%llu B): %s", (unsigned long long)nbytes, hipGetErrorString(e)); }
    if ((e = hipMalloc((void**)&s->d_norm, (size_t)std::max<uint32_t>(n_docs, 1) * 4)) != hipSuccess) { cleanup(); return fail(ctx, NS_E_NOMEM, "hipMalloc norm: %s", hipGetErrorString(e)); }
    if ((e = hipMalloc((void**)&s->d_pnorm, nbytes / 2 + kPadPostings * 4)) != hipSuccess) { cleanup(); return fail(ctx, NS_E_NOMEM, "hipMalloc per-posting norms (%llu B): %s", (unsigned long long)(nbytes / 2), hipGetErrorString(e)); }
    if ((e = hipMalloc((void**)&s->d_len, (size_t)std::max<uint32_t>(n_docs, 1) * 4)) != hipSuccess) { cleanup(); return fail(ctx, NS_E_NOMEM, "hipMalloc doc_len: %s", hipGetErrorString(e)); }
    for (int i = 0; i < 2; i++)
        if (hipHostMalloc(&s->pin[i], kStageChunk, hipHostMallocDefault) != hipSuccess || hipEventCreateWithFlags(&s->ev[i], hipEventDisableTiming) != hipSuccess) {
            cleanup();
            return fail(ctx, NS_E_NOMEM, "pinned staging allocation failed");
        }
    e = hipMemsetAsync((char*)s->d_postings + nbytes, 0xFF, kPadPostings * 8, ctx->stream);   // docId ~0: never taken
    if (e == hipSuccess) e = hipMemsetAsync((char*)s->d_pnorm + nbytes / 2, 0, kPadPostings * 4, ctx->stream);
    if (e == hipSuccess && n_docs) e = seg_stage(ctx, s, s->d_len, doc_len, (size_t)n_docs * 4);
    if (e == hipSuccess && n_docs) {
        hipLaunchKernelGGL(k_norm, dim3((n_docs + 255) / 256), dim3(256), 0, ctx->stream, s->d_len, s->d_norm, n_docs, avgdl);
        e = hipGetLastError();
    }
    if (e == hipSuccess && n_docs) {
        // The packed stream's norm plane: a posting carries a 16-bit index into the table of the segment's DISTINCT norms
        // (norm is a function of doc_len alone) instead of the fp32 norm — no quantisation, 2 B instead of 4.  The table is
        // k_norm over the distinct lengths: the same expression on the same inputs, the same bits.  Segments with more
        // than 65536 distinct lengths get no table (and no packed stream).
        uint32_t max_len = 0;
        for (uint32_t i = 0; i < n_docs; i++) max_len = std::max(max_len, doc_len[i]);
        std::vector<uint32_t> uniq;
        std::vector<uint16_t> nidx;
        if (max_len < (1u << 24)) {
            std::vector<uint32_t> rank((size_t)max_len + 1, 0u);
            for (uint32_t i = 0; i < n_docs; i++) rank[doc_len[i]] = 1u;
            for (uint32_t v = 0; v <= max_len; v++)
                if (rank[v]) { rank[v] = (uint32_t)uniq.size(); uniq.push_back(v); }
            if (uniq.size() <= 65536) {
                nidx.resize(n_docs);
                for (uint32_t i = 0; i < n_docs; i++) nidx[i] = (uint16_t)rank[doc_len[i]];
            }
        } else {
            uniq.assign(doc_len, doc_len + n_docs);
            std::sort(uniq.begin(), uniq.end());
            uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
            if (uniq.size() <= 65536) {
                nidx.resize(n_docs);
                for (uint32_t i = 0; i < n_docs; i++) nidx[i] = (uint16_t)(std::lower_bound(uniq.begin(), uniq.end(), doc_len[i]) - uniq.begin());
            }
        }
        if (!nidx.empty()) {
            uint32_t* d_ulen = nullptr;
            const size_t D = uniq.size();
            e = hipMalloc((void**)&s->d_nidx, (size_t)n_docs * 2);
            if (e == hipSuccess) e = hipMalloc((void**)&s->d_ntab, D * 4);
            if (e == hipSuccess) e = hipMalloc((void**)&d_ulen, D * 4);
            if (e == hipSuccess) e = seg_stage(ctx, s, s->d_nidx, nidx.data(), (size_t)n_docs * 2);
            if (e == hipSuccess) e = seg_stage(ctx, s, d_ulen, uniq.data(), D * 4);
            if (e == hipSuccess) {
                hipLaunchKernelGGL(k_norm, dim3((uint32_t)((D + 255) / 256)), dim3(256), 0, ctx->stream, d_ulen, s->d_ntab, (uint32_t)D, avgdl);
                e = hipGetLastError();
            }
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);   // d_ulen and the host vectors die here
            (void)hipFree(d_ulen);
            s->n_norms = (uint32_t)D;
        }
    }
    if (e != hipSuccess) { (void)hipStreamSynchronize(ctx->stream); cleanup(); return fail(ctx, NS_E_HIP, "segment upload: %s", hipGetErrorString(e)); }
    {
        // norms are monotone in doc_len (k_norm's expression, evaluated here the same way): the extremes bound them all
        uint32_t dl_min = 0xFFFFFFFFu, dl_max = 0;
        for (uint32_t i = 0; i < n_docs; i++) { dl_min = std::min(dl_min, doc_len[i]); dl_max = std::max(dl_max, doc_len[i]); }
        auto norm_of = [&](uint32_t dl) { return 1.2f * ((1.0f - 0.75f) + 0.75f * ((float)dl / avgdl)); };
        const float lo_ok = 9.5367431640625e-07f /* 2^-20 */, hi_ok = 1073741824.0f /* 2^30 */;
        s->norm_safe = n_docs > 0 && std::isfinite(avgdl) && avgdl > 0.0f && norm_of(dl_min) >= lo_ok && norm_of(dl_min) <= hi_ok &&
                       norm_of(dl_max) >= lo_ok && norm_of(dl_max) <= hi_ok;
    }
    ctx->pending_uploads.push_back(s);
    *out = s;
    return NS_OK;
}

extern "C" int ns_segment_upload_append(ns_ctx* ctx, ns_seg* s, const void* bytes, uint64_t nbytes) {
    if (!ctx || !s || s->ctx != ctx || !s->pending) return fail(ctx, NS_E_STATE, "ns_segment_upload_append: no upload in progress for this segment");
    if (nbytes % 8 != 0) return fail(ctx, NS_E_INVAL, "chunk of %llu bytes is not a whole number of {u32,u32} pairs", (unsigned long long)nbytes);
    if (s->filled + nbytes > s->n_postings * 8) return fail(ctx, NS_E_INVAL, "chunk runs past the %llu payload bytes announced at begin", (unsigned long long)(s->n_postings * 8));
    if (nbytes && !bytes) return fail(ctx, NS_E_INVAL, "null doc_len/postings");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const hipError_t e = seg_stage(ctx, s, (char*)s->d_postings + s->filled, bytes, nbytes);
    if (e != hipSuccess) return fail(ctx, NS_E_HIP, "segment upload: %s", hipGetErrorString(e));
    s->filled += nbytes;
    return NS_OK;
}

extern "C" int ns_segment_upload_end(ns_ctx* ctx, ns_seg* s) {
    if (!ctx || !s || s->ctx != ctx || !s->pending) return fail(ctx, NS_E_STATE, "ns_segment_upload_end: no upload in progress for this segment");
    if (s->filled != s->n_postings * 8) return fail(ctx, NS_E_STATE, "ns_segment_upload_end: %llu of %llu payload bytes received", (unsigned long long)s->filled, (unsigned long long)(s->n_postings * 8));
    if (s->id < ctx->segs.size() && ctx->segs[s->id]) return fail(ctx, NS_E_INVAL, "segment %u already uploaded", s->id);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipError_t e = hipSuccess;
    if (s->n_postings) {
        uint32_t blocks = (uint32_t)std::min<uint64_t>((s->n_postings + 255) / 256, 65536);
        hipLaunchKernelGGL(k_pnorm, dim3(blocks), dim3(256), 0, ctx->stream, s->d_postings, s->d_norm, s->d_pnorm, s->n_postings, s->n_docs);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return fail(ctx, NS_E_HIP, "segment upload: %s", hipGetErrorString(e));
    seg_free_staging(s);
    s->pending = false;
    ctx->pending_uploads.erase(std::remove(ctx->pending_uploads.begin(), ctx->pending_uploads.end(), s), ctx->pending_uploads.end());
    if (ctx->segs.size() <= s->id) ctx->segs.resize(s->id + 1, nullptr);
    ctx->segs[s->id] = s;
    return NS_OK;
}

extern "C" int ns_segment_upload(ns_ctx* ctx, uint32_t seg_id, uint32_t n_docs, float avgdl, const uint32_t* doc_len,
                                 const void* postings, uint64_t nbytes, ns_seg** out) {
    if (out) *out = nullptr;
    if (ctx && nbytes && !postings) return fail(ctx, NS_E_INVAL, "null doc_len/postings");
    ns_seg* s = nullptr;
    int rc = ns_segment_upload_begin(ctx, seg_id, n_docs, avgdl, doc_len, nbytes, &s);
    if (rc != NS_OK) return rc;
    rc = ns_segment_upload_append(ctx, s, postings, nbytes);
    if (rc == NS_OK) rc = ns_segment_upload_end(ctx, s);
    if (rc != NS_OK) { const std::string keep = ctx->err; (void)ns_segment_release(ctx, s); ctx->err = keep; return rc; }
    if (out) *out = s;
    return NS_OK;
}

extern "C" int ns_segment_release(ns_ctx* ctx, ns_seg* seg) {
    if (!ctx || !seg) return fail(ctx, NS_E_INVAL, "ns_segment_release: null argument");
    if (seg->ctx != ctx) return fail(ctx, NS_E_INVAL, "segment does not belong to this ctx");
    if (!seg->pending && (seg->id >= ctx->segs.size() || ctx->segs[seg->id] != seg)) return fail(ctx, NS_E_INVAL, "segment does not belong to this ctx");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->alt_stream) HIPCHK(ctx, hipStreamSynchronize(ctx->alt_stream));
    if (ctx->pull_stream) HIPCHK(ctx, hipStreamSynchronize(ctx->pull_stream));
    seg_free_staging(seg);
    seg_free_device(seg);
    // the shared-score registry forgets the segment's lists: a later segment under the same id starts with an empty interval map
    // and must see every one of its lists as new (that is when overlaps are checked)
    if (!seg->pending && ctx->share_n) {
        std::vector<ns_ctx::ShareEnt> kept(ctx->share_tab.size());
        const size_t m = kept.size() - 1;
        size_t n = 0;
        for (const auto& en : ctx->share_tab)
            if (en.key != ~0ull && (uint32_t)(en.key >> 32) != seg->id) {
                size_t h = (size_t)((en.key * 0x9E3779B97F4A7C15ull) >> 20) & m;
                while (kept[h].key != ~0ull) h = (h + 1) & m;
                kept[h] = en; n++;
            }
        ctx->share_tab.swap(kept);
        ctx->share_n = n;
    }
    if (!seg->pending) ctx->segs[seg->id] = nullptr;
    else ctx->pending_uploads.erase(std::remove(ctx->pending_uploads.begin(), ctx->pending_uploads.end(), seg), ctx->pending_uploads.end());
    delete seg;
    return NS_OK;
}

__global__ void __launch_bounds__(256) k_pk_scores(const uint2* __restrict__ impacts, float* __restrict__ scores, uint64_t n) {
    for (uint64_t p = (uint64_t)blockIdx.x * 256 + threadIdx.x; p < n; p += (uint64_t)gridDim.x * 256) scores[p] = __uint_as_float(impacts[p].y);
}
// the packed form of the impact stream = the scores alone, per posting (the docIds come from the packed doc plane)
static hipError_t seg_fill_pk_scores(ns_ctx* ctx, ns_seg* seg) {
    if (!seg->d_packed || !seg->d_impacts || !seg->n_postings) return hipSuccess;
    const uint64_t n = seg->n_postings + kPadPostings;
    hipError_t e = hipSuccess;
    if (!seg->d_pk_scores) e = hipMalloc((void**)&seg->d_pk_scores, n * 4);
    if (e != hipSuccess) { seg->d_pk_scores = nullptr; return e; }
    hipLaunchKernelGGL(k_pk_scores, dim3((uint32_t)std::min<uint64_t>((n + 255) / 256, 1u << 16)), dim3(256), 0, ctx->stream, seg->d_impacts, seg->d_pk_scores, n);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    return e;
}

// One thread per posting: the list that holds it is found by binary search over the (sorted) list starts.
// The arithmetic is src/api_engine.cpp:477-479 operation for operation with the compiler's IEEE division —
// the expression the scoring kernels evaluate per posting per query when no impact stream exists.
__global__ void __launch_bounds__(256) k_build_impacts(const uint2* __restrict__ postings, const float* __restrict__ pnorm,
                                                       uint2* __restrict__ impacts, const uint32_t* __restrict__ starts,
                                                       const uint32_t* __restrict__ counts, const float* __restrict__ idfs,
                                                       uint32_t n_lists, uint64_t n_postings) {
    for (uint64_t p = (uint64_t)blockIdx.x * 256 + threadIdx.x; p < n_postings; p += (uint64_t)gridDim.x * 256) {
        uint32_t lo = 0, hi = n_lists;
        while (lo < hi) {
            const uint32_t mid = lo + ((hi - lo) >> 1);
            if ((uint64_t)starts[mid] <= p) lo = mid + 1; else hi = mid;
        }
        if (lo == 0) continue;
        const uint32_t l = lo - 1;
        if (p - starts[l] >= counts[l]) continue;
        const uint2 pv = postings[p];
        const float tf = (float)pv.y;
        const float den = tf + pnorm[p];
        const float num = idfs[l] * (tf * (1.2f + 1.0f));
        impacts[p] = make_uint2(pv.x, __float_as_uint(num / den));
    }
}

extern "C" int ns_segment_build_impacts(ns_ctx* ctx, ns_seg* seg, const uint64_t* byte_off, const uint32_t* counts,
                                        const float* idfs, uint32_t n_lists) {
    if (!ctx || !seg) return fail(ctx, NS_E_INVAL, "ns_segment_build_impacts: null argument");
    if (seg->id >= ctx->segs.size() || ctx->segs[seg->id] != seg) return fail(ctx, NS_E_INVAL, "segment does not belong to this ctx");
    if (n_lists && (!byte_off || !counts || !idfs)) return fail(ctx, NS_E_INVAL, "null list arrays");
    if (!n_lists || !seg->n_postings) return NS_OK;
    if (ctx->live_shared) return fail(ctx, NS_E_STATE, "ns_segment_build_impacts: %u batch(es) that compute shared term scores into the same buffer are alive; destroy them first", ctx->live_shared);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    struct L { uint32_t first, count; float idf; };
    std::vector<L> lists;
    lists.reserve(n_lists);
    for (uint32_t i = 0; i < n_lists; i++) {
        if (byte_off[i] % 8 != 0) return fail(ctx, NS_E_INVAL, "list %u: byte offset %llu is not a multiple of 8", i, (unsigned long long)byte_off[i]);
        const uint64_t first = byte_off[i] / 8;
        if (first + counts[i] > seg->n_postings) return fail(ctx, NS_E_INVAL, "list %u runs past the segment's postings", i);
        if (counts[i]) lists.push_back({(uint32_t)first, counts[i], idfs[i]});
    }
    std::sort(lists.begin(), lists.end(), [](const L& a, const L& b) { return a.first < b.first; });
    for (size_t i = 1; i < lists.size(); i++)
        if (lists[i - 1].first + (uint64_t)lists[i - 1].count > lists[i].first) return fail(ctx, NS_E_INVAL, "lists overlap at posting %u", lists[i].first);
    if (lists.empty()) return NS_OK;
    hipError_t e = hipSuccess;
    if (!seg->d_impacts) {
        e = hipMalloc((void**)&seg->d_impacts, (seg->n_postings + kPadPostings) * 8);
        if (e != hipSuccess) { seg->d_impacts = nullptr; return fail(ctx, NS_E_NOMEM, "hipMalloc impact stream (%llu B): %s", (unsigned long long)(seg->n_postings * 8), hipGetErrorString(e)); }
        e = hipMemsetAsync(seg->d_impacts, 0xFF, (seg->n_postings + kPadPostings) * 8, ctx->stream);   // docId ~0: never taken
    }
    const size_t n = lists.size();
    std::vector<uint32_t> h_starts(n), h_counts(n);
    std::vector<float> h_idfs(n);
    for (size_t i = 0; i < n; i++) { h_starts[i] = lists[i].first; h_counts[i] = lists[i].count; h_idfs[i] = lists[i].idf; }
    uint32_t* d_tmp = nullptr;
    if (e == hipSuccess) e = hipMalloc((void**)&d_tmp, n * 12);
    if (e == hipSuccess) e = hipMemcpyAsync(d_tmp, h_starts.data(), n * 4, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_tmp + n, h_counts.data(), n * 4, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_tmp + 2 * n, h_idfs.data(), n * 4, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        const uint32_t blocks = (uint32_t)std::min<uint64_t>((seg->n_postings + 255) / 256, 1u << 16);
        hipLaunchKernelGGL(k_build_impacts, dim3(blocks), dim3(256), 0, ctx->stream, seg->d_postings, seg->d_pnorm, seg->d_impacts,
                           d_tmp, d_tmp + n, (const float*)(d_tmp + 2 * n), (uint32_t)n, seg->n_postings);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_tmp);
    if (e == hipSuccess) e = seg_fill_pk_scores(ctx, seg);
    if (e != hipSuccess) return fail(ctx, NS_E_HIP, "ns_segment_build_impacts: %s", hipGetErrorString(e));
    // registry: old entries + new ones (a list given again replaces its entry)
    std::vector<ns_seg::ImpList> all;
    for (const auto& t : seg->imp_tab) if (t.first != 0xFFFFFFFFu) all.push_back(t);
    for (const auto& l : lists) { ns_seg::ImpList t; t.first = l.first; t.count = l.count; std::memcpy(&t.idf_bits, &l.idf, 4); all.push_back(t); }
    size_t cap = 16;
    while (cap < all.size() * 2) cap <<= 1;
    std::vector<ns_seg::ImpList> tab(cap);
    size_t distinct = 0;
    for (const auto& t : all) {
        for (size_t h = ((size_t)t.first * 0x9E3779B1u) & (cap - 1);; h = (h + 1) & (cap - 1)) {
            if (tab[h].first == 0xFFFFFFFFu) { tab[h] = t; distinct++; break; }
            if (tab[h].first == t.first) { tab[h] = t; break; }
        }
    }
    seg->imp_tab.swap(tab);
    seg->imp_lists = distinct;
    return NS_OK;
}

// ---- shared term scores -------------------------------------------------------------------------------------------
// A batch names the same posting list many times (cfg5: 16384 queries draw ~50 000 term refs from ~40 000 distinct lists, and
// the 32 hot lists ~460 times each): the BM25 term score of a posting, src/api_engine.cpp:477-479, depends on the list and on
// the list's idf, not on the query.  A sharing batch therefore computes the scores of every DISTINCT list it names once, in
// this kernel, in front of its scoring kernel and on every run (nothing is kept from one batch for the next: a later batch
// finds the buffer as if it had never been written), and the scoring bodies read {docId, score bits} — their IMP form, the one
// that serves the optional impact stream — instead of {docId, tf} + norm.  Same operations, same order, same bits: the
// division below is the compiler's correctly rounded one, which ns_div_short reproduces exactly where it is used.
// 1024 consecutive postings of the batch's build order per workgroup; a thread finds the list of each of its postings by
// binary search between the lists that hold the workgroup's first and last posting.
__global__ void __launch_bounds__(256) k_share_scores(const DevShare* __restrict__ sh, uint32_t n_lists, const DevSeg* __restrict__ segs) {
    const uint32_t total = sh[n_lists].before;
    const uint32_t base = blockIdx.x * 1024u;
    if (base >= total) return;
    const uint32_t last = min(base + 1023u, total - 1u);
    __shared__ uint32_t s_lo, s_hi;
    if (threadIdx.x < 2) {
        const uint32_t want = threadIdx.x == 0 ? base : last;   // the list l with sh[l].before <= want < sh[l + 1].before
        uint32_t lo = 0, hi = n_lists;
        while (hi - lo > 1) {
            const uint32_t mid = lo + ((hi - lo) >> 1);
            if (sh[mid].before <= want) lo = mid; else hi = mid;
        }
        if (threadIdx.x == 0) s_lo = lo; else s_hi = lo;
    }
    __syncthreads();
    const uint32_t l0 = s_lo, l1 = s_hi;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t p = base + (uint32_t)j * 256u + threadIdx.x;
        if (p > last) continue;
        uint32_t lo = l0, hi = l1 + 1u;
        while (hi - lo > 1) {
            const uint32_t mid = lo + ((hi - lo) >> 1);
            if (sh[mid].before <= p) lo = mid; else hi = mid;
        }
        const DevShare L = sh[lo];
        const DevSeg& sg = segs[L.seg];
        const uint64_t at = (uint64_t)L.first + (p - L.before);
        const uint2 pv = sg.postings[at];
        const float tf = (float)pv.y;
        const float den = tf + sg.pnorm[at];
        const float num = L.idf * (tf * (1.2f + 1.0f));
        const_cast<uint2*>(sg.impacts)[at] = make_uint2(pv.x, __float_as_uint(num / den));
    }
}

extern "C" int ns_ctx_share_scores(ns_ctx* ctx, int mode) {
    if (!ctx) return fail(nullptr, NS_E_INVAL, "ns_ctx_share_scores: ctx is NULL");
    if (mode < 0 || mode > 2) return fail(ctx, NS_E_INVAL, "ns_ctx_share_scores: mode %d outside [0, 2]", mode);
    ctx->share_mode = mode;
    return NS_OK;
}

// the registry entry of list (seg, first): found or made (the table doubles at half load; entries are never removed)
static ns_ctx::ShareEnt& share_entry(ns_ctx* ctx, uint32_t seg, uint32_t first, bool& fresh) {
    if (ctx->share_tab.empty() || ctx->share_n * 2 >= ctx->share_tab.size()) {
        std::vector<ns_ctx::ShareEnt> bigger(std::max<size_t>(1024, ctx->share_tab.size() * 2));
        const size_t m = bigger.size() - 1;
        for (const auto& e : ctx->share_tab)
            if (e.key != ~0ull) {
                size_t h = (size_t)((e.key * 0x9E3779B97F4A7C15ull) >> 20) & m;
                while (bigger[h].key != ~0ull) h = (h + 1) & m;
                bigger[h] = e;
            }
        ctx->share_tab.swap(bigger);
    }
    const uint64_t key = ((uint64_t)seg << 32) | first;
    const size_t m = ctx->share_tab.size() - 1;
    size_t h = (size_t)((key * 0x9E3779B97F4A7C15ull) >> 20) & m;
    for (;; h = (h + 1) & m) {
        ns_ctx::ShareEnt& e = ctx->share_tab[h];
        if (e.key == key) { fresh = false; return e; }
        if (e.key == ~0ull) { e.key = key; ctx->share_n++; fresh = true; return e; }
    }
}

extern "C" int ns_segment_build_packed(ns_ctx* ctx, ns_seg* seg) {
    if (!ctx || !seg) return fail(ctx, NS_E_INVAL, "ns_segment_build_packed: null argument");
    if (seg->pending || seg->id >= ctx->segs.size() || ctx->segs[seg->id] != seg) return fail(ctx, NS_E_INVAL, "segment does not belong to this ctx");
    if (seg->d_packed || !seg->n_postings) return NS_OK;
    if (!seg->d_nidx) return fail(ctx, NS_E_INVAL, "ns_segment_build_packed: segment %u has more than 65536 distinct document lengths; its norms do not fit a 16-bit index", seg->id);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const uint64_t n_blocks = (seg->n_postings + kPkBlock - 1) / kPkBlock;
    hipError_t e = hipMalloc((void**)&seg->d_packed, (n_blocks + 2) * kPkStrideDwords * 4);   // + slack: a round may be planned one block past the end
    if (e == hipSuccess) e = hipMalloc((void**)&seg->d_pk_hdr, (n_blocks + 2) * sizeof(uint2));
    if (e == hipSuccess) e = hipMemsetAsync(seg->d_packed + n_blocks * kPkStrideDwords, 0, 2 * kPkStrideDwords * 4, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(seg->d_pk_hdr + n_blocks, 0, 2 * sizeof(uint2), ctx->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_pack, dim3((uint32_t)n_blocks), dim3(64), 0, ctx->stream, seg->d_postings, seg->d_nidx, seg->d_packed, seg->d_pk_hdr,
                           seg->n_postings, seg->n_docs, (uint32_t)n_blocks);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess) e = seg_fill_pk_scores(ctx, seg);
    if (e != hipSuccess) {
        (void)hipFree(seg->d_packed); (void)hipFree(seg->d_pk_hdr);
        seg->d_packed = nullptr; seg->d_pk_hdr = nullptr;
        return fail(ctx, e == hipErrorOutOfMemory ? NS_E_NOMEM : NS_E_HIP, "ns_segment_build_packed: %s", hipGetErrorString(e));
    }
    return NS_OK;
}

// ---- skip tables (block metadata next to the reference's posting format) --------------------------------
// entry[l][i] = index of list l's first posting with docId >= i * kSkipDocs.  k_skip_fill: every entry = the list's end;
// k_skip_build: one thread per posting p of a registered list — the grid cells after the previous posting's, up to its own,
// start at p (docIds ascend inside a list, so every cell is written at most once); k_skip_check: a table is usable when
// it starts at the list's first posting, never decreases and ends at the list's end (an unsorted list fails here and
// simply keeps no table: its groups take the cursor path, which tolerates any order).
__global__ void __launch_bounds__(256) k_skip_fill(uint32_t* __restrict__ sk, const uint32_t* __restrict__ starts, const uint32_t* __restrict__ counts,
                                                   uint32_t per_list) {
    const uint32_t l = blockIdx.y;
    const uint32_t endp = starts[l] + counts[l];
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < per_list; i += gridDim.x * 256) sk[(uint64_t)l * per_list + i] = endp;
}
__global__ void __launch_bounds__(256) k_skip_build(const uint2* __restrict__ postings, uint32_t* __restrict__ sk, const uint32_t* __restrict__ starts,
                                                    const uint32_t* __restrict__ counts, uint32_t per_list) {
    const uint32_t l = blockIdx.y;
    const uint32_t first = starts[l], n = counts[l];
    uint32_t* row = sk + (uint64_t)l * per_list;
    const uint32_t cells = per_list - 2;   // entries 0 .. cells are cell starts (cells = one past the last doc's), + 1 spare
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const uint32_t d = postings[first + i].x / kSkipDocs;
        uint32_t c0 = 0;
        if (i) {
            const uint32_t dp = postings[first + i - 1].x / kSkipDocs;
            c0 = dp + 1;
        }
        for (uint32_t c = c0; c <= d && c <= cells; c++) row[c] = first + i;
    }
}
__global__ void __launch_bounds__(256) k_skip_check(const uint2* __restrict__ postings, const uint32_t* __restrict__ sk, const uint32_t* __restrict__ starts,
                                                    const uint32_t* __restrict__ counts, uint32_t per_list, uint32_t* __restrict__ bad) {
    const uint32_t l = blockIdx.y;
    const uint32_t first = starts[l], endp = first + counts[l];
    const uint32_t* row = sk + (uint64_t)l * per_list;
    bool b = false;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i + 1 < per_list; i += gridDim.x * 256) {
        const uint32_t a = row[i], z = row[i + 1];
        if (a > z || a < first || z > endp) b = true;
        if (i == 0 && a != first) b = true;
        if (i + 2 == per_list && (z != endp || a != endp)) b = true;
        // the postings of cell i really are inside it (first and last suffice: the list ascends between table entries
        // or the next cell's check fails)
        if (i + 2 < per_list && a < z) {
            if (postings[a].x / kSkipDocs != i || postings[z - 1].x / kSkipDocs != i) b = true;
        }
    }
    if (b) bad[l] = 1u;
}

extern "C" int ns_segment_build_skips(ns_ctx* ctx, ns_seg* seg, const uint64_t* byte_off, const uint32_t* counts, uint32_t n_lists) {
    if (!ctx || !seg) return fail(ctx, NS_E_INVAL, "ns_segment_build_skips: null argument");
    if (seg->pending || seg->id >= ctx->segs.size() || ctx->segs[seg->id] != seg) return fail(ctx, NS_E_INVAL, "segment does not belong to this ctx");
    if (n_lists && (!byte_off || !counts)) return fail(ctx, NS_E_INVAL, "null list arrays");
    if (!n_lists || !seg->n_postings || !seg->n_docs) return NS_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    struct L { uint32_t first, count; };
    std::vector<L> lists;
    lists.reserve(n_lists);
    for (uint32_t i = 0; i < n_lists; i++) {
        if (byte_off[i] % 8 != 0) return fail(ctx, NS_E_INVAL, "list %u: byte offset %llu is not a multiple of 8", i, (unsigned long long)byte_off[i]);
        const uint64_t first = byte_off[i] / 8;
        if (first + counts[i] > seg->n_postings) return fail(ctx, NS_E_INVAL, "list %u runs past the segment's postings", i);
        if (counts[i] && !seg->skip_of((uint32_t)first, counts[i])) lists.push_back({(uint32_t)first, counts[i]});
    }
    std::sort(lists.begin(), lists.end(), [](const L& a, const L& b) { return a.first < b.first || (a.first == b.first && a.count < b.count); });
    lists.erase(std::unique(lists.begin(), lists.end(), [](const L& a, const L& b) { return a.first == b.first; }), lists.end());
    if (lists.empty()) return NS_OK;
    const uint32_t per_list = seg->skip_entries();
    const size_t n = lists.size();
    const uint64_t need = seg->skip_used + (uint64_t)n * per_list;
    if (need >= (1ull << 32) - 1) return fail(ctx, NS_E_INVAL, "ns_segment_build_skips: %llu table entries do not fit 32 bits", (unsigned long long)need);
    hipError_t e = hipSuccess;
    if (need > seg->skip_cap) {   // grow (tables already built are kept: batches prepared earlier hold indices into them)
        uint32_t* bigger = nullptr;
        const uint64_t cap = std::max<uint64_t>(need, seg->skip_cap * 2);
        e = hipMalloc((void**)&bigger, cap * 4);
        if (e != hipSuccess) return fail(ctx, NS_E_NOMEM, "hipMalloc skip tables (%llu B): %s", (unsigned long long)(cap * 4), hipGetErrorString(e));
        if (seg->d_skips) {   // the old block stays until the segment goes: earlier batches point into it
            e = hipMemcpyAsync(bigger, seg->d_skips, seg->skip_used * 4, hipMemcpyDeviceToDevice, ctx->stream);
            seg->skip_retired.push_back(seg->d_skips);
        }
        seg->d_skips = bigger;
        seg->skip_cap = cap;
        if (e != hipSuccess) return fail(ctx, NS_E_HIP, "ns_segment_build_skips: %s", hipGetErrorString(e));
    }
    std::vector<uint32_t> h(n * 3, 0u);
    for (size_t i = 0; i < n; i++) { h[i] = lists[i].first; h[n + i] = lists[i].count; }
    uint32_t* d_tmp = nullptr;
    e = hipMalloc((void**)&d_tmp, n * 12);
    if (e == hipSuccess) e = hipMemcpyAsync(d_tmp, h.data(), n * 12, hipMemcpyHostToDevice, ctx->stream);
    uint32_t* base = seg->d_skips + seg->skip_used;
    if (e == hipSuccess) {
        const dim3 g1((per_list + 255) / 256 > 64 ? 64 : (per_list + 255) / 256, (uint32_t)n);
        hipLaunchKernelGGL(k_skip_fill, g1, dim3(256), 0, ctx->stream, base, d_tmp, d_tmp + n, per_list);
        uint32_t cmax = 0;
        for (const auto& l : lists) cmax = std::max(cmax, l.count);
        const dim3 g2(std::min<uint32_t>((cmax + 255) / 256, 1024u), (uint32_t)n);
        hipLaunchKernelGGL(k_skip_build, g2, dim3(256), 0, ctx->stream, seg->d_postings, base, d_tmp, d_tmp + n, per_list);
        hipLaunchKernelGGL(k_skip_check, g1, dim3(256), 0, ctx->stream, seg->d_postings, base, d_tmp, d_tmp + n, per_list, d_tmp + 2 * n);
        e = hipGetLastError();
    }
    std::vector<uint32_t> bad(n, 1u);
    if (e == hipSuccess) e = hipMemcpyAsync(bad.data(), d_tmp + 2 * n, n * 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_tmp);
    if (e != hipSuccess) return fail(ctx, NS_E_HIP, "ns_segment_build_skips: %s", hipGetErrorString(e));
    // registry
    std::vector<ns_seg::SkipList> all;
    for (const auto& t : seg->skip_tab) if (t.first != 0xFFFFFFFFu) all.push_back(t);
    for (size_t i = 0; i < n; i++) {
        if (bad[i]) continue;   // not ascending: no table
        ns_seg::SkipList t; t.first = lists[i].first; t.count = lists[i].count; t.entry = (uint32_t)(seg->skip_used + i * per_list);
        all.push_back(t);
    }
    seg->skip_used = need;
    size_t cap = 16;
    while (cap < all.size() * 2) cap <<= 1;
    std::vector<ns_seg::SkipList> tab(cap);
    for (const auto& t : all)
        for (size_t hh = ((size_t)t.first * 0x9E3779B1u) & (cap - 1);; hh = (hh + 1) & (cap - 1))
            if (tab[hh].first == 0xFFFFFFFFu) { tab[hh] = t; break; }
    seg->skip_tab.swap(tab);
    seg->skip_lists = all.size();
    return NS_OK;
}

// Block maxima of the given lists (DevSeg::blockmax; ns_prune_kernel.hip): per 256 postings of a list the largest BM25 term
// score with the given idf.  Lists given again (same first posting) are rebuilt with the new idf; tables already built stay
// where they are (batches prepared earlier hold indices into them).
extern "C" int ns_segment_build_blockmax(ns_ctx* ctx, ns_seg* seg, const uint64_t* byte_off, const uint32_t* counts, const float* idfs, uint32_t n_lists) {
    if (!ctx || !seg) return fail(ctx, NS_E_INVAL, "ns_segment_build_blockmax: null argument");
    if (seg->pending || seg->id >= ctx->segs.size() || ctx->segs[seg->id] != seg) return fail(ctx, NS_E_INVAL, "segment does not belong to this ctx");
    if (n_lists && (!byte_off || !counts || !idfs)) return fail(ctx, NS_E_INVAL, "null list arrays");
    if (!n_lists || !seg->n_postings) return NS_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    struct L { uint32_t first, count; float idf; uint32_t entry; };
    std::vector<L> lists;
    lists.reserve(n_lists);
    for (uint32_t i = 0; i < n_lists; i++) {
        if (byte_off[i] % 8 != 0) return fail(ctx, NS_E_INVAL, "list %u: byte offset %llu is not a multiple of 8", i, (unsigned long long)byte_off[i]);
        const uint64_t first = byte_off[i] / 8;
        if (first + counts[i] > seg->n_postings) return fail(ctx, NS_E_INVAL, "list %u runs past the segment's postings", i);
        uint32_t ib; std::memcpy(&ib, &idfs[i], 4);
        if (counts[i] && !seg->bmx_of((uint32_t)first, counts[i], ib)) lists.push_back({(uint32_t)first, counts[i], idfs[i], 0u});
    }
    std::sort(lists.begin(), lists.end(), [](const L& a, const L& b) { return a.first < b.first; });
    lists.erase(std::unique(lists.begin(), lists.end(), [](const L& a, const L& b) { return a.first == b.first; }), lists.end());
    if (lists.empty()) return NS_OK;
    uint64_t need = seg->bmx_used;
    uint32_t bmax_blocks = 0;
    for (auto& l : lists) {
        const uint32_t nb = (l.count + kBmxBlock - 1) / kBmxBlock;
        l.entry = (uint32_t)need;
        need += nb;
        bmax_blocks = std::max(bmax_blocks, nb);
    }
    if (need + 64 >= (1ull << 32) - 1) return fail(ctx, NS_E_INVAL, "ns_segment_build_blockmax: %llu entries do not fit 32 bits", (unsigned long long)need);
    hipError_t e = hipSuccess;
    if (need + 64 > seg->bmx_cap) {   // +64: a wave reads 64 maxima at a time, possibly past a list's last block
        float* bigger = nullptr;
        const uint64_t cap = std::max<uint64_t>(need + 64, seg->bmx_cap * 2);
        e = hipMalloc((void**)&bigger, cap * 4);
        if (e != hipSuccess) return fail(ctx, NS_E_NOMEM, "hipMalloc block maxima (%llu B): %s", (unsigned long long)(cap * 4), hipGetErrorString(e));
        e = hipMemsetAsync(bigger, 0, cap * 4, ctx->stream);
        if (e == hipSuccess && seg->d_blockmax) {
            e = hipMemcpyAsync(bigger, seg->d_blockmax, seg->bmx_used * 4, hipMemcpyDeviceToDevice, ctx->stream);
            seg->bmx_retired.push_back(seg->d_blockmax);
        }
        seg->d_blockmax = bigger;
        seg->bmx_cap = cap;
        if (e != hipSuccess) return fail(ctx, NS_E_HIP, "ns_segment_build_blockmax: %s", hipGetErrorString(e));
    }
    const size_t n = lists.size();
    std::vector<uint32_t> h(n * 4);
    for (size_t i = 0; i < n; i++) {
        h[i] = lists[i].first; h[n + i] = lists[i].count;
        std::memcpy(&h[2 * n + i], &lists[i].idf, 4);
        h[3 * n + i] = lists[i].entry;
    }
    uint32_t* d_tmp = nullptr;
    e = hipMalloc((void**)&d_tmp, n * 16);
    if (e == hipSuccess) e = hipMemcpyAsync(d_tmp, h.data(), n * 16, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        for (size_t l0 = 0; l0 < n; l0 += 32768) {   // grid.y is limited to 65535
            const uint32_t ny = (uint32_t)std::min<size_t>(32768, n - l0);
            hipLaunchKernelGGL(k_blockmax, dim3(std::min<uint32_t>(std::max<uint32_t>(bmax_blocks, 1u), 256u), ny), dim3(64), 0, ctx->stream,
                               seg->d_postings, seg->d_pnorm, seg->d_blockmax, d_tmp + l0, d_tmp + n + l0, (const float*)(d_tmp + 2 * n + l0), d_tmp + 3 * n + l0);
        }
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_tmp);
    if (e != hipSuccess) return fail(ctx, NS_E_HIP, "ns_segment_build_blockmax: %s", hipGetErrorString(e));
    seg->bmx_used = need;
    std::vector<ns_seg::BmxList> all;
    for (const auto& t : seg->bmx_tab) if (t.first != 0xFFFFFFFFu) all.push_back(t);
    for (const auto& l : lists) { ns_seg::BmxList t; t.first = l.first; t.count = l.count; std::memcpy(&t.idf_bits, &l.idf, 4); t.entry = l.entry; all.push_back(t); }
    size_t cap = 16;
    while (cap < all.size() * 2) cap <<= 1;
    std::vector<ns_seg::BmxList> tab(cap);
    for (const auto& t : all)
        for (size_t hh = ((size_t)t.first * 0x9E3779B1u) & (cap - 1);; hh = (hh + 1) & (cap - 1)) {
            if (tab[hh].first == 0xFFFFFFFFu || tab[hh].first == t.first) { tab[hh] = t; break; }   // a list given again: the later table wins
        }
    seg->bmx_tab.swap(tab);
    return NS_OK;
}

extern "C" int ns_ctx_use_merge(ns_ctx* ctx, int on) {
    if (!ctx) return NS_E_INVAL;
    ctx->use_merge = on != 0;
    return NS_OK;
}

extern "C" int ns_ctx_use_pruning(ns_ctx* ctx, int on) {
    if (!ctx) return NS_E_INVAL;
    ctx->use_pruning = on != 0;
    return NS_OK;
}

extern "C" int ns_ctx_use_skips(ns_ctx* ctx, int on) {
    if (!ctx) return NS_E_INVAL;
    ctx->use_skips = on != 0;
    return NS_OK;
}

extern "C" int ns_ctx_use_packed(ns_ctx* ctx, int on) {
    if (!ctx) return NS_E_INVAL;
    if (on < 0 || on > 2) return fail(ctx, NS_E_INVAL, "ns_ctx_use_packed: mode %d (0, 1 or 2)", on);
    ctx->use_packed = on;
    return NS_OK;
}

extern "C" int ns_ctx_set_overlap(ns_ctx* ctx, int on) {
    if (!ctx) return NS_E_INVAL;
    if (on && !ctx->alt_stream) return fail(ctx, NS_E_HIP, "ns_ctx_set_overlap: the second stream could not be created");
    ctx->overlap = on != 0;
    return NS_OK;
}

extern "C" int ns_ctx_set_host_threads(ns_ctx* ctx, uint32_t n) {
    if (!ctx) return NS_E_INVAL;
    if (n > 64) return fail(ctx, NS_E_INVAL, "ns_ctx_set_host_threads: %u threads (at most 64)", n);
    ctx->prep_threads = n;
    return NS_OK;
}

extern "C" int ns_ctx_use_impacts(ns_ctx* ctx, int on) {
    if (!ctx) return NS_E_INVAL;
    ctx->use_impacts = on != 0;
    return NS_OK;
}

// ------------------------------------------------------------------------------------------------
struct ns_batch {
    ns_ctx* ctx = nullptr;
    uint32_t Q = 0, K = 0, flags = 0;
    uint32_t variant = 0, tile_docs = 0, hb = 0;
    uint32_t n_items = 0;      // workgroup-kernel items (k_score)
    uint32_t n_witems = 0;     // wave-kernel items
    uint32_t n_class[3] = {0, 0, 0};   // auto mode: items of class S (thin foreign lists), D (dense foreign), T (very dense: doc tiles); contiguous in d_witems
    uint32_t n_bgroups = 0;    // term groups that need the boundary prepass (k_score path only)
    uint32_t n_terms = 0, n_parts = 0;
    uint64_t postings = 0;
    bool direct = false;   // every query has exactly one work item: the scoring kernel writes final rows
    int pk = 0;            // every segment of the batch has a packed posting stream and the ctx wants it: 1 = packed docIds + tf, norms from the fp32 norm stream; 2 = norms through the 16-bit norm index (ns_ctx_use_packed)
    bool pruned = false;   // some single-term items take the block-max pruned body (ns_ctx_use_pruning)
    bool imp = false;      // every list of the batch has an impact stream: the kernels read {docId, score} instead of {docId, tf} + norm
    bool shared = false;   // ... because the batch computes them itself, once per distinct list and run (k_share_scores; ns_ctx_share_scores)
    uint32_t n_share = 0;          // distinct lists the batch builds
    uint64_t share_postings = 0;   // their postings
    DevShare* d_share = nullptr;   // n_share + 1 entries
    // device
    DevItem* d_items = nullptr;
    DevWItem* d_witems = nullptr;
    DevTerm* d_terms = nullptr;
    DevGroup* d_groups = nullptr;
    DevQuery* d_queries = nullptr;
    DevSeg* d_segs = nullptr;
    uint32_t* d_bounds = nullptr;
    Hit* d_part_hits = nullptr;
    uint32_t* d_part_nhits = nullptr;
    uint64_t* d_part_found = nullptr;
    uint32_t* d_heads = nullptr;
    Hit* d_hits = nullptr;
    uint32_t* d_nhits = nullptr;
    uint64_t* d_found = nullptr;
    // active output pointers (own or bound)
    Hit* o_hits = nullptr;
    uint32_t* o_nhits = nullptr;
    uint64_t* o_found = nullptr;
    // timed runs: four events per run (start, before scoring, after scoring, end), read back at sync
    std::vector<hipEvent_t> ev_pool;
    size_t ev_pending = 0;          // events recorded since the last sync
    bool ran = false;
    float last_score_ms = -1.0f, last_total_ms = -1.0f;
    double sum_score_ms = 0.0, sum_total_ms = 0.0;
    uint32_t timed_runs = 0;
    uint32_t* d_wide_q = nullptr;
    uint32_t n_wide_q = 0;
    size_t out_span = 0, off_nhits = 0, off_found = 0;   // [d_hits .. d_found end) is one contiguous span of the block
    std::vector<std::pair<void*, size_t>> blocks;   // every device block of this batch (returned to the ctx pool on destroy)
    // pipelined use (NS_RUN_FETCH): the results' D2H copy into a pinned slot of the ctx is enqueued right behind the
    // kernels and `done` is recorded after it, so that fetch / destroy wait for THIS batch only, not for the stream
    hipStream_t st = nullptr;  // the stream all of this batch's work goes to (the ctx's, or its second one when overlap is on)
    hipEvent_t done = nullptr;
    bool done_recorded = false;
    bool fetch_enqueued = false;   // the last run carried NS_RUN_FETCH
    int down_slot = -1;        // index into ns_ctx::down_slots while a D2H copy is pending or unread
};

static constexpr size_t kPoolMaxBytes = 1ull << 30;   // cached blocks beyond this are released
static constexpr size_t kStageMaxBytes = 256ull << 20;   // larger batches upload/fetch array by array
static constexpr size_t kPullUploadBytes = 256 << 10;     // uploads up to this size are pulled by a kernel instead of the DMA engine
static constexpr size_t kHostResultBytes = 64 << 10;      // result arrays up to this size live in pinned host memory

// Block sizes are quantised (powers of two from 64 KB to 64 MB, multiples of 16 MB above) so that a serving loop's
// batches of slightly different shape reuse each other's blocks instead of going to hipMalloc.
static size_t pool_quantise(size_t n) {
    if (n <= (64u << 10)) return 64u << 10;
    if (n <= (64u << 20)) { size_t q = 64u << 10; while (q < n) q <<= 1; return q; }
    return (n + (16u << 20) - 1) & ~(size_t)((16u << 20) - 1);
}
static hipError_t pool_alloc(ns_ctx* ctx, void** out, size_t n) {
    n = pool_quantise(n);
    size_t best = (size_t)-1;
    for (size_t i = 0; i < ctx->pool.size(); i++)
        if (ctx->pool[i].n >= n && ctx->pool[i].n <= 2 * n + 4096 && (best == (size_t)-1 || ctx->pool[i].n < ctx->pool[best].n)) best = i;
    if (best != (size_t)-1) {
        *out = ctx->pool[best].p;
        ctx->pool_bytes -= ctx->pool[best].n;
        ctx->pool[best] = ctx->pool.back();
        ctx->pool.pop_back();
        return hipSuccess;
    }
    hipError_t e = hipMalloc(out, n);
    if (e == hipErrorOutOfMemory && !ctx->pool.empty()) {   // give the cache back and try once more
        for (auto& b : ctx->pool) (void)hipFree(b.p);
        ctx->pool.clear(); ctx->pool_bytes = 0;
        e = hipMalloc(out, n);
    }
    return e;
}
// caller has synchronised the stream: nothing in flight uses the block
static void pool_free(ns_ctx* ctx, void* p, size_t n) {
    if (!p) return;
    n = pool_quantise(n);
    if (ctx->pool_bytes + n > kPoolMaxBytes || ctx->pool.size() >= 64) { (void)hipFree(p); return; }
    ctx->pool.push_back({p, n});
    ctx->pool_bytes += n;
}

__global__ void __launch_bounds__(256) k_pull(uint4* __restrict__ dst, const uint4* __restrict__ src, uint32_t n16) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n16) dst[i] = src[i];
}

extern "C" void ns_batch_destroy(ns_batch* b) {
    if (!b) return;
    (void)hipSetDevice(b->ctx->device);
    // nothing of THIS batch may still be in flight when its blocks go back to the pool; later batches on the same
    // stream are none of its business (a pipelined caller destroys batch i while batch i+1 runs)
    if (b->done_recorded) (void)hipEventSynchronize(b->done);
    else (void)hipStreamSynchronize(b->st);
    if (b->ctx->down_owner == b) b->ctx->down_owner = nullptr;
    if (b->down_slot >= 0) b->ctx->down_slots[(size_t)b->down_slot].busy = false;
    if (b->shared && b->ctx->live_shared) b->ctx->live_shared--;
    for (auto& blk : b->blocks) pool_free(b->ctx, blk.first, blk.second);
    for (auto& e : b->ev_pool) if (e) (void)hipEventDestroy(e);
    if (b->done) (void)hipEventDestroy(b->done);
    delete b;
}

static hipError_t batch_alloc(ns_batch* b, void** dptr, size_t n) {
    hipError_t e = pool_alloc(b->ctx, dptr, n);
    if (e == hipSuccess) b->blocks.push_back({*dptr, n});
    return e;
}
// ---- host side of a batch: term refs -> (query, segment) groups -> work items, on several host threads ----------
// A batch is prepared in three fork-join phases over contiguous slices of the queries (the reference's requests are
// independent, src/api_engine.cpp:369): A regroup + classify, B cut into work items, C write the descriptors into the
// pinned staging buffer in launch order.  Between the phases only prefix sums over the slices run serially.  Every
// result (descriptor bytes, launch order) is independent of the number of threads.
namespace {

struct HostGroup { DevGroup g; uint32_t query; uint64_t cost; uint64_t cmax; uint64_t work; bool wave; uint8_t cls; bool fast_div; bool signed_in; bool grid; bool merge2; };

constexpr uint32_t kOrderBuckets = 2048;   // launch-order key: 6 bits of exponent x 5 bits of mantissa of the estimated run time
inline uint32_t order_bucket(uint64_t c) {  // descending: bucket 0 holds the longest items
    if (c < 32) return kOrderBuckets - 1 - (uint32_t)c;
    const int b = 63 - __builtin_clzll(c);
    const uint32_t key = (uint32_t)b * 32u + (uint32_t)((c >> (b - 5)) & 31u);
    return kOrderBuckets - 1 - std::min(key, kOrderBuckets - 1);
}

struct PrepSlice {
    uint32_t q0 = 0, q1 = 0;
    std::vector<DevTerm> dterms;
    std::vector<HostGroup> groups;
    std::vector<uint32_t> qgroup_begin;   // q1 - q0 + 1 entries, local group indices
    std::vector<uint32_t> seg_ids;
    uint64_t bounds_total = 0, postings_total = 0, total_work = 0;
    bool all_imp = true, all_pk = true, any_pruned = false;
    int err_code = NS_OK;
    uint32_t err_query = 0xFFFFFFFFu;
    std::string err_msg;
    // phase B
    std::vector<DevWItem> witems;
    std::vector<uint16_t> wbucket;        // launch-order bucket of each wave item; bit 15: > 16 terms (the "wide" instantiation)
    std::vector<uint32_t> wshare;         // locality key of the item: segment (6 bits) | doc range on the 4096-grid (12) | hash of its largest list (14) -> XCD dealing
    std::vector<DevItem> items;
    std::vector<uint64_t> item_cost;
    std::vector<DevGroup> bgroups;
    uint32_t n_rows = 0;
    bool direct = true;
    std::vector<uint32_t> hist;           // [2][kOrderBuckets]: narrow, wide
    // offsets handed down by the serial steps
    uint32_t term_off = 0, row_off = 0, item_off = 0, bgroup_off = 0;
    uint64_t bounds_off = 0;
    std::vector<uint32_t> start;          // [2][kOrderBuckets]: this slice's first position in each bucket of the sorted item array
    void reset(uint32_t a, uint32_t b) {
        q0 = a; q1 = b;
        dterms.clear(); groups.clear(); qgroup_begin.clear(); seg_ids.clear();
        bounds_total = postings_total = total_work = 0; all_imp = true; all_pk = true; any_pruned = false;
        err_code = NS_OK; err_query = 0xFFFFFFFFu; err_msg.clear();
        witems.clear(); wbucket.clear(); wshare.clear(); items.clear(); item_cost.clear(); bgroups.clear();
        n_rows = 0; direct = true;
        hist.assign(2 * kOrderBuckets, 0u);
        start.assign(2 * kOrderBuckets, 0u);
        term_off = row_off = item_off = bgroup_off = 0; bounds_off = 0;
    }
    void fail_at(uint32_t q, int code, const char* fmt, ...) {
        if (err_code != NS_OK) return;   // the first failing query of the slice is reported
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        std::vsnprintf(buf, sizeof(buf), fmt, ap);
        va_end(ap);
        err_code = code; err_query = q; err_msg = buf;
    }
};

}  // namespace

struct ns_prep {
    std::vector<PrepSlice> slices;
    std::vector<DevShare> share_build;    // shared term scores: the distinct lists the batch builds, in build order
    std::vector<uint32_t> share_at;       // per launch position: the item's locality key (XCD dealing)
    std::vector<uint32_t> bucket_pos;     // launch position at which each fine bucket of the narrow half starts (+ the end)
    std::vector<std::vector<DevWItem>> deal_tmp;   // per host thread
    std::vector<std::vector<uint64_t>> deal_key, deal_alt;
    std::vector<std::vector<uint32_t>> deal_bins;
    ForkJoin* pool = nullptr;
    ~ns_prep() { delete pool; }
};
static void prep_free(ns_prep* p) { delete p; }

extern "C" int ns_batch_prepare(ns_ctx* ctx, const ns_query_desc* queries, const ns_term_ref* terms, uint32_t n_queries,
                                uint32_t k, uint32_t flags, ns_batch** out) {
    if (!ctx) return fail(nullptr, NS_E_INVAL, "ns_batch_prepare: ctx is NULL");
    if (!out) return fail(ctx, NS_E_INVAL, "ns_batch_prepare: out is NULL");
    *out = nullptr;
    if (k < 1 || k > NS_MAX_K) return fail(ctx, NS_E_INVAL, "k=%u outside [1,%u]", k, NS_MAX_K);
    if (n_queries && !queries) return fail(ctx, NS_E_INVAL, "queries is NULL");
    if (flags & ~NS_FLAG_AND) return fail(ctx, NS_E_INVAL, "unknown flags 0x%x", flags);
    HIPCHK(ctx, hipSetDevice(ctx->device));

    const VariantDesc vd = kVariants[ctx->variant];
    const uint32_t tile_docs = vd.nt * vd.spt;
    const bool wave_path = vd.hb != 0;
    const bool auto_mode = ctx->variant == 0;

    // per-batch segment table (n_tiles depends on the workgroup-kernel variant)
    std::vector<DevSeg> segs(ctx->segs.size());
    for (size_t i = 0; i < ctx->segs.size(); i++) {
        DevSeg d{};
        if (ns_seg* s = ctx->segs[i]) {
            d.postings = s->d_postings;
            d.pnorm = s->d_pnorm;
            d.impacts = s->d_impacts;
            d.packed = s->d_packed;
            d.pk_hdr = s->d_pk_hdr;
            d.ntab = s->d_ntab;
            d.pk_scores = s->d_pk_scores;
            d.skips = s->d_skips;
            d.blockmax = s->d_blockmax;
            d.norm = s->d_norm;
            d.n_postings = s->n_postings;
            d.n_docs = s->n_docs;
            d.n_tiles = (s->n_docs + tile_docs - 1) / tile_docs;
        }
        segs[i] = d;
    }

    // ---- slices: one per host thread for large batches (below ~1500 queries per thread the hand-over costs more than it saves) ----
    if (!ctx->prep) ctx->prep = new ns_prep();
    ns_prep& P = *ctx->prep;
    unsigned width = 1;
    if (n_queries >= 3000 && ctx->prep_threads != 1) {
        unsigned want = ctx->prep_threads ? ctx->prep_threads : std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 8u);
        // one thread per ~1500 queries or ~6000 term refs, whichever asks for more (a query over 8 segments carries 8x the refs)
        uint64_t n_refs = 0;
        for (uint32_t q = 0; q < n_queries; q++) n_refs += queries[q].term_count;
        width = std::max(1u, std::min<unsigned>(want, std::max<unsigned>(n_queries / 1500, (unsigned)std::min<uint64_t>(n_refs / 6000, 64))));
    }
    if (width > 1 && (!P.pool || P.pool->width() < width)) { delete P.pool; P.pool = new ForkJoin(width); }
    if (P.slices.size() < width) P.slices.resize(width);
    for (unsigned s = 0; s < width; s++) P.slices[s].reset((uint32_t)((uint64_t)n_queries * s / width), (uint32_t)((uint64_t)n_queries * (s + 1) / width));
    auto fork = [&](const std::function<void(unsigned)>& fn) {
        if (width == 1) fn(0); else P.pool->run(width, fn);
    };
    const bool want_imp = ctx->use_impacts && auto_mode;

    // ---- phase A: regroup term refs by (query, segment), keeping query-term order inside each group; classify ----
    fork([&](unsigned si) {
        PrepSlice& S = P.slices[si];
        S.all_imp = want_imp;   // stays true while every list met so far has an impact stream built with this idf
        S.qgroup_begin.reserve(S.q1 - S.q0 + 1);
        for (uint32_t q = S.q0; q < S.q1; q++) {
            S.qgroup_begin.push_back((uint32_t)S.groups.size());
            const ns_query_desc qd = queries[q];
            if (qd.term_count && !terms) { S.fail_at(q, NS_E_INVAL, "terms is NULL"); break; }
            S.seg_ids.clear();
            bool bad = false;
            for (uint32_t i = 0; i < qd.term_count && !bad; i++) {
                const ns_term_ref& r = terms[qd.term_begin + i];
                if (r.seg_id >= ctx->segs.size() || !ctx->segs[r.seg_id]) { S.fail_at(q, NS_E_INVAL, "query %u term %u: unknown segment %u", q, i, r.seg_id); bad = true; break; }
                const ns_seg* s = ctx->segs[r.seg_id];
                if (r.byte_off % 8 != 0) { S.fail_at(q, NS_E_INVAL, "query %u term %u: byte_off %llu not a multiple of 8", q, i, (unsigned long long)r.byte_off); bad = true; break; }
                if (r.byte_off / 8 + r.count > s->n_postings) { S.fail_at(q, NS_E_INVAL, "query %u term %u: list [%llu,+%u) outside segment %u (%llu postings)", q, i, (unsigned long long)(r.byte_off / 8), r.count, r.seg_id, (unsigned long long)s->n_postings); bad = true; break; }
                if (std::find(S.seg_ids.begin(), S.seg_ids.end(), r.seg_id) == S.seg_ids.end()) S.seg_ids.push_back(r.seg_id);
            }
            if (bad) break;
            std::sort(S.seg_ids.begin(), S.seg_ids.end());   // segments in manifest (id) order, api_engine.cpp:441
            for (uint32_t sid : S.seg_ids) {
                HostGroup hg{};
                hg.fast_div = ctx->segs[sid]->norm_safe;
                if (!ctx->segs[sid]->d_packed) S.all_pk = false;
                hg.g.term_begin = (uint32_t)S.dterms.size();   // local to the slice until phase B
                hg.g.seg = sid;
                hg.query = q;
                for (uint32_t i = 0; i < qd.term_count; i++) {
                    const ns_term_ref& r = terms[qd.term_begin + i];
                    if (r.seg_id != sid) continue;
                    DevTerm t{};
                    t.list_off = r.byte_off / 8;
                    t.count = r.count;
                    t.idf = r.idf;
                    t.weight = r.qweight;
                    t.seg = sid;
                    S.dterms.push_back(t);
                    hg.cost += r.count;
                    hg.cmax = std::max<uint64_t>(hg.cmax, r.count);
                    if (S.all_imp) {
                        uint32_t ib; std::memcpy(&ib, &r.idf, 4);
                        S.all_imp = ctx->segs[sid]->imp_has((uint32_t)(r.byte_off / 8), r.count, ib);
                    }
                    // 2^-30 <= idf <= 2^30 (and finite): see ns_div_short
                    if (!(r.idf >= 9.313225746154785e-10f && r.idf <= 1073741824.0f)) hg.fast_div = false;
                    if (std::signbit(r.idf) || std::signbit(r.qweight)) hg.signed_in = true;   // a contribution may be -0.0f (see dscore_body)
                }
                hg.g.term_count = (uint32_t)S.dterms.size() - hg.g.term_begin;
                if ((flags & NS_FLAG_AND) && hg.g.term_count > 255) { S.fail_at(q, NS_E_INVAL, "AND mode supports at most 255 term refs per (query, segment)"); bad = true; break; }
                hg.wave = wave_path && hg.g.term_count <= kWaveMaxTerms;
                // class of the group (auto mode only): which scoring body suits its mix of lists (sweeps on
                // MI355X, profiles/r01): 1 = one list dominates (the others hold <= 1/32 of its postings): driver
                // stream with the 64-posting foreign budget; 2 = dense (>= 0.25 postings per doc over >= 2
                // lists): doc tiles; 0 = driver stream with the 192-posting budget.
                {
                    const uint64_t rest = hg.cost - hg.cmax;
                    const uint32_t nd = std::max<uint32_t>(segs[sid].n_docs, 1);
                    if (rest * 32 <= hg.cmax) hg.cls = 1;
                    else if (hg.g.term_count >= 2 && hg.cost * 64 >= (uint64_t)nd * ctx->tile_dens64) hg.cls = 2;
                    else hg.cls = 0;
                    // work estimate in units of one streamed driver posting (measured, profiles/r01): a foreign
                    // posting (claim, accumulate, read back) costs ~8x, a doc-tile posting ~2x
                    hg.work = !auto_mode ? hg.cost : (hg.cls == 2 ? hg.cost * kWorkTile : hg.cmax + rest * kWorkForeign);
                    // two lists, general class: the merge body (no table): both lists cost about alike per posting
                    // (two COMPARABLE lists: when one is more than 8x the other, a window of the short one per round of the long
                    // one is mostly padding and the table path is as good: r8 + r300, 37 : 1, measured 3 % slower with the merge)
                    hg.merge2 = auto_mode && ctx->use_merge && hg.cls == 0 && hg.g.term_count == 2 && rest * ctx->merge_ratio >= hg.cmax;
                    if (hg.merge2) hg.work = hg.cmax + rest * kWorkMerge;
                }
                if (!hg.wave) {
                    hg.g.bounds_off = S.bounds_total;   // local; the slice's base is added in phase B
                    S.bounds_total += (uint64_t)(segs[sid].n_tiles + 1) * hg.g.term_count;
                }
                S.postings_total += hg.cost;
                S.total_work += hg.work;
                S.groups.push_back(hg);
            }
            if (bad) break;
        }
        S.qgroup_begin.resize(S.q1 - S.q0 + 1, (uint32_t)S.groups.size());
    });
    {
        const PrepSlice* first = nullptr;
        for (unsigned s = 0; s < width; s++)
            if (P.slices[s].err_code != NS_OK && (!first || P.slices[s].err_query < first->err_query)) first = &P.slices[s];
        if (first) return fail(ctx, first->err_code, "%s", first->err_msg.c_str());
    }
    uint64_t bounds_total = 0, postings_total = 0, total_work = 0;
    uint32_t n_dterms = 0, G = 0;
    bool all_imp = want_imp, all_pk = ctx->use_packed != 0 && auto_mode;
    for (unsigned s = 0; s < width; s++) {
        PrepSlice& S = P.slices[s];
        S.term_off = n_dterms; S.bounds_off = bounds_total;
        n_dterms += (uint32_t)S.dterms.size(); G += (uint32_t)S.groups.size();
        bounds_total += S.bounds_total; postings_total += S.postings_total; total_work += S.total_work;
        all_imp = all_imp && S.all_imp;
        all_pk = all_pk && S.all_pk;
    }
    if (bounds_total >= (1ull << 32)) return fail(ctx, NS_E_INVAL, "batch too large: %llu boundary entries; split the batch", (unsigned long long)bounds_total);

    // ---- shared term scores (ns_ctx_share_scores; k_share_scores): the batch's distinct lists, each listed once, in the
    // order the term refs name them.  A list is refused — and the batch then scores every posting in place, as it
    // always did — when it overlaps another list ever shared in its segment (two builders, one posting), when the segment
    // carries an optional impact stream that does not hold exactly this list with this idf (the stream is not the batch's
    // to overwrite), or when its idf differs from the one a LIVE sharing batch built it with (that batch may run again).
    std::vector<DevShare>& share_build = P.share_build;
    share_build.clear();
    uint64_t share_postings = 0;
    bool shared = false;
    if (!all_imp && want_imp && ctx->share_mode != 0 && !all_pk && postings_total > 0 &&
        (ctx->share_mode == 2 || postings_total >= ctx->share_min_postings)) {
        shared = true;
        if (++ctx->share_epoch == 0) {   // the batch counter wrapped: no entry may look like this batch's
            for (auto& en : ctx->share_tab) en.epoch = 0;
            ctx->share_epoch = 1;
        }
        const uint32_t epoch = ctx->share_epoch;
        for (unsigned sl = 0; sl < width && shared; sl++) {
            const std::vector<DevTerm>& dts = P.slices[sl].dterms;
            for (size_t ti = 0; ti < dts.size(); ti++) {
                // the registry is a few MB and every probe of it a cache miss: the probe of the term 8 ahead is requested now
                if (ti + 8 < dts.size() && !ctx->share_tab.empty()) {
                    const uint64_t k8 = ((uint64_t)dts[ti + 8].seg << 32) | (uint32_t)dts[ti + 8].list_off;
                    __builtin_prefetch(&ctx->share_tab[(size_t)((k8 * 0x9E3779B97F4A7C15ull) >> 20) & (ctx->share_tab.size() - 1)]);
                }
                const DevTerm& t = dts[ti];
                if (!t.count) continue;
                ns_seg* sg = ctx->segs[t.seg];
                uint32_t ib; std::memcpy(&ib, &t.idf, 4);
                const uint32_t first = (uint32_t)t.list_off;
                if (sg->imp_lists) {
                    if (sg->imp_has(first, t.count, ib)) continue;   // the optional stream holds this list already
                    shared = false; break;
                }
                bool fresh = false;
                ns_ctx::ShareEnt& en = share_entry(ctx, t.seg, first, fresh);
                if (fresh) { en.count = t.count; en.idf_bits = ib; en.epoch = 0; en.bad = !sg->share_admit(first, t.count); }
                if (en.bad || en.count != t.count) { shared = false; break; }
                if (en.idf_bits != ib) {
                    if (ctx->live_shared || en.epoch == epoch) { shared = false; break; }
                    en.idf_bits = ib;
                }
                if (en.epoch != epoch) {
                    en.epoch = epoch;
                    if (share_postings + t.count >= (1ull << 32)) { shared = false; break; }
                    share_build.push_back(DevShare{first, t.count, t.idf, t.seg, (uint32_t)share_postings});
                    share_postings += t.count;
                    // (a batch that cannot reach the ratio gives up here: the frequent lists come early, and with them the verdict)
                    if (ctx->share_mode == 1 && share_postings * ctx->share_ratio > postings_total) { shared = false; break; }
                }
            }
        }
        if (shared && ctx->share_mode == 1 && postings_total < (uint64_t)ctx->share_ratio * share_postings) shared = false;
        for (size_t i = 0; shared && i < share_build.size(); i++) {   // the score buffers of the segments the batch builds into
            ns_seg* sg = ctx->segs[share_build[i].seg];
            if (!sg->d_impacts) {
                const size_t nb = (size_t)(sg->n_postings + kPadPostings) * 8;
                hipError_t ea = hipMalloc((void**)&sg->d_impacts, nb);
                if (ea == hipSuccess) ea = hipMemsetAsync(sg->d_impacts, 0xFF, nb, ctx->stream);   // docId ~0: never taken
                if (ea == hipSuccess) ea = hipStreamSynchronize(ctx->stream);
                if (ea != hipSuccess) {   // no room for the scores: the batch scores in place
                    if (sg->d_impacts) (void)hipFree(sg->d_impacts);
                    sg->d_impacts = nullptr; (void)hipGetLastError();
                    shared = false;
                }
            }
            if (shared) segs[share_build[i].seg].impacts = sg->d_impacts;
        }
        if (shared) all_imp = true;
        else { share_build.clear(); share_postings = 0; }
    }

    // ---- work items.  A group is split into doc ranges (a) so that no single worker carries more
    // than ~split_postings units of estimated work (the longest item bounds the batch's tail; launch
    // order is longest-estimated-work first), and (b) so that a small batch still fills the chip.  Partial rows of one query are contiguous; k_merge joins them.
    const uint32_t min_items = ctx->min_items ? ctx->min_items : (uint32_t)std::max(ctx->n_cus, 1) * 24u;
    int small_mode = 0;   // thin / tile items: 0 = double share, 1 = plain share, 2 = half share (see below)
    uint64_t split_postings = ctx->split_postings ? ctx->split_postings
                              : (!auto_mode ? kDefaultSplitPostings : (k <= 32 ? kSplitWorkSmallK : kSplitWorkLargeK));
    bool fine_cut = false;   // the batch is cut finer than the default share: it does not fill the chip for long
    if (!ctx->split_postings && auto_mode) {
        // a small batch: cut finer so that the chip still sees ~100 items per CU (an item of the default size
        // runs 0.3-1.3 ms: with fewer items than wave slots that would be the whole batch's time), but not
        // below ~16 K units, where an item's fixed cost takes over
        const uint64_t fine = total_work / ((uint64_t)std::max(ctx->n_cus, 1) * 96u);
        fine_cut = fine < split_postings;
        split_postings = std::min<uint64_t>(split_postings, std::max<uint64_t>(fine, 16384));
        // A batch that leaves wave slots idle is bound by its LONGEST item, and a streaming item is a chain of dependent
        // round trips (one 256-posting round in flight per wave).  When the double share of a thin or tile item would
        // exceed what a wave slot gets on average, those items lose it; when even a plain share does, they are halved
        // (profiles/r02/small_batch_split_modes.txt: 256 / 512 / 1024 queries of the cfg5 law run 36 / 27 / 14 % faster;
        // batches that fill the chip — all thin groups of cfg5 alone, 4096 single-term queries — are left as they were).
        const uint64_t per_slot = total_work / ((uint64_t)std::max(ctx->n_cus, 1) * 24u);
        small_mode = per_slot >= 2 * split_postings ? 0 : (per_slot >= split_postings ? 1 : 2);
    }
    uint32_t chunks_per_group = 1;
    if (G > 0 && G < min_items) chunks_per_group = std::min<uint32_t>((min_items + G - 1) / G, 1024u);   // one query alone: 1024 ranges are plenty
    std::vector<DevQuery> dq(n_queries);

    // ---- phase B: cut the groups into work items (rows numbered inside the slice) ----
    fork([&](unsigned si) {
        PrepSlice& S = P.slices[si];
        for (uint32_t q = S.q0; q < S.q1; q++) {
            dq[q].part_begin = S.n_rows;
            for (uint32_t gi = S.qgroup_begin[q - S.q0]; gi < S.qgroup_begin[q - S.q0 + 1]; gi++) {
                HostGroup& hg = S.groups[gi];
                hg.g.term_begin += S.term_off;          // global from here on
                const DevSeg& sg = segs[hg.g.seg];
                if (sg.n_docs == 0) continue;   // empty segment: nothing to score
                if (hg.wave) {
                    // thin and tile groups run at a steady rate per posting: fewer, longer items (less per-item set-up,
                    // same balance); groups with dense foreign lists vary more per posting and stay finer
                    uint64_t sp_ = (auto_mode && hg.cls != 0) ? split_postings * 2 : split_postings;
                    if (auto_mode && hg.cls != 0 && small_mode) sp_ = small_mode == 2 ? split_postings / 2 : split_postings;
                    const uint64_t want = std::max<uint64_t>((hg.work + sp_ - 1) / sp_, chunks_per_group);
                    uint32_t ns = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(want, 1), std::min<uint32_t>(sg.n_docs, 4096));
                    // The number of ranges is a POWER OF TWO (the nearest in ratio; the next one up for a small batch): range boundaries then come from one
                    // nested grid, so the items of different queries cover IDENTICAL doc ranges of the lists they share, and
                    // items of one range — equal size, adjacent in the launch order — read the same bytes at about the same
                    // time: L2 / Infinity-Cache hits instead of misses (same number of items on average; cfg3 -5 %, cfg5's
                    // tile groups -7 %, a 2048-query batch -4 %; profiles/r02/ab/ab14_range_grid.txt).
                    if (auto_mode && ns > 1) {
                        uint32_t p2 = 1;
                        while (p2 < ns) p2 <<= 1;
                        if (!small_mode && (uint64_t)ns * ns * 2 < (uint64_t)p2 * p2) p2 >>= 1;   // a batch that leaves wave slots idle never gets fewer items
                        // K > 32: a group that is cut at all is cut into at least 8 ranges (cfg3 -7 %, at K = 64 -7 %, another seed
                        // -7 %; 16 is too many; at K <= 32 the same rule costs 2 %: ab14 / ab19)
                        if (k > 32 && p2 < 8) p2 = 8;
                        ns = std::min<uint32_t>(p2, std::min<uint32_t>(sg.n_docs, 4096));
                    }
                    // Skip tables (ns_segment_build_skips): a doc-tile group walks the grid of its lists' tables; a group of the
                    // driver-stream bodies that is cut into ranges takes the ranges' ends of its frequent lists from their
                    // tables instead of searching for them (the searches of a hot list are a dozen dependent loads: 4-20 % of
                    // an item's time, most in small batches).  Either way the ranges start and end on the grid.
                    if (auto_mode && ctx->use_skips && (hg.cls == 2 || ns > 1) && !ctx->segs[hg.g.seg]->skip_tab.empty()) {
                        const ns_seg* sg_ = ctx->segs[hg.g.seg];
                        DevTerm* dt = S.dterms.data() + (hg.g.term_begin - S.term_off);
                        for (uint32_t ti = 0; ti < hg.g.term_count; ti++) {
                            if (dt[ti].count < kSkipMinCount) continue;
                            dt[ti].skip = sg_->skip_of((uint32_t)dt[ti].list_off, dt[ti].count);
                            if (dt[ti].skip) hg.grid = true;
                        }
                    }
                    // Block-max pruning (ns_ctx_use_pruning): a group of ONE list whose block maxima were built with this idf, scored
                    // with a positive weight — the fp32 product w * s is then monotone in s and never a negative zero
                    bool pruned = false;
                    if (auto_mode && ctx->use_pruning && hg.g.term_count == 1 && !ctx->segs[hg.g.seg]->bmx_tab.empty()) {
                        DevTerm* dt = S.dterms.data() + (hg.g.term_begin - S.term_off);
                        uint32_t ib; std::memcpy(&ib, &dt[0].idf, 4);
                        if (dt[0].idf > 0.0f && std::isfinite(dt[0].idf) && dt[0].weight > 0.0f && std::isfinite(dt[0].weight)) {
                            dt[0].bmx = ctx->segs[hg.g.seg]->bmx_of((uint32_t)dt[0].list_off, dt[0].count, ib);
                            pruned = dt[0].bmx != 0;
                        }
                        if (pruned) S.any_pruned = true;
                    }
                    // launch-order key = estimated run time of the ITEM: its share of the group's work plus what
                    // every item pays per term regardless of size (window planning, range searches, table set-up)
                    const uint64_t per_term = hg.cls == 2 ? kItemTermTile : (hg.cls == 1 ? kItemTermThin : kItemTermGeneral);
                    uint64_t key = hg.work / ns + 1 + (auto_mode ? per_term * hg.g.term_count : 0);
                    // In a batch cut finer than the default share the streaming items (thin, tile) are chains of round trips that
                    // a less loaded chip does not shorten, while the issue-bound general items do run faster: those start
                    // later (2048 queries of the cfg5 law -11 %, 4096 -5 %, 512 -4 %; ab16_order_key_small_batches.txt).
                    if (fine_cut && hg.cls == 0) key = key * 5 / 8;
                    // K > 32: the streaming items pay more per posting than the work units (fitted at K = 10) say — fewer waves
                    // per CU, larger candidate buffers — so the general items start later there too (cfg3 -2.6 %, at K = 64 -2.3 %)
                    else if (auto_mode && k > 32 && hg.cls == 0) key = key * 3 / 4;
                    if (auto_mode) key = key * ctx->key_pct[hg.merge2 ? 3 : hg.cls] / 100;   // sweeps (NS_KEY_PCT); 100 each by default
                    const bool wide = auto_mode && hg.g.term_count > 16;
                    const uint32_t bucket = order_bucket(key);
                    // what the group's items read most of: its largest list (the driver of a driver-stream item)
                    uint64_t big_off = 0;
                    {
                        const DevTerm* dt = S.dterms.data() + (hg.g.term_begin - S.term_off);
                        uint32_t cm = 0;
                        for (uint32_t ti = 0; ti < hg.g.term_count; ti++)
                            if (dt[ti].count >= cm) { cm = dt[ti].count; big_off = dt[ti].list_off; }
                    }
                    for (uint32_t i = 0; i < ns; i++) {
                        DevWItem it{};
                        it.query = q;
                        it.seg = hg.g.seg;
                        it.term_begin = hg.g.term_begin;
                        it.term_count = hg.g.term_count;
                        it.doc_lo = (uint32_t)((uint64_t)sg.n_docs * i / ns);
                        it.doc_hi = (uint32_t)((uint64_t)sg.n_docs * (i + 1) / ns);
                        if (hg.grid) {   // ranges of a skip-grid group start and end on the grid
                            it.doc_lo -= it.doc_lo % kSkipDocs;
                            if (i + 1 < ns) it.doc_hi -= it.doc_hi % kSkipDocs;
                        }
                        if (it.doc_hi <= it.doc_lo) continue;
                        it.out_slot = S.n_rows++;
                        it.whole = (ns == 1 ? 1u : 0u) | (hg.fast_div ? 8u : 0u) | (hg.signed_in ? 16u : 0u) | (hg.grid ? (hg.cls == 2 ? 32u : 64u) : 0u);
                        // auto mode: very dense groups take the doc-tile body (bit 1), groups with thin non-driver lists the small foreign budget (bit 2)
                        if (auto_mode) it.whole |= (hg.cls == 2 ? 2u : 0u) | (hg.cls == 1 ? 4u : 0u) | (pruned ? 128u : 0u) | (hg.merge2 && hg.wave ? 256u : 0u);
                        S.witems.push_back(it);
                        S.wbucket.push_back((uint16_t)(bucket | (wide ? 0x8000u : 0u)));
                        {
                            uint64_t h = big_off * 0x9E3779B97F4A7C15ull;
                            h ^= h >> 29;
                            const uint32_t r12 = (uint32_t)(((uint64_t)it.doc_lo << 12) / std::max<uint32_t>(sg.n_docs, 1u)) & 4095u;
                            S.wshare.push_back(((it.seg & 63u) << 26) | (r12 << 14) | (uint32_t)((h >> 40) & 0x3FFFu));
                        }
                        S.hist[(wide ? kOrderBuckets : 0) + bucket]++;
                    }
                } else {
                    DevGroup g = hg.g;
                    g.bounds_off += S.bounds_off;
                    S.bgroups.push_back(g);
                    const uint32_t nt = sg.n_tiles;
                    const uint32_t chunks = std::min(chunks_per_group, nt);
                    const uint32_t per = (nt + chunks - 1) / chunks;
                    for (uint32_t tb = 0; tb < nt; tb += per) {
                        DevItem it{};
                        it.bounds_off = g.bounds_off;
                        it.query = q;
                        it.seg = g.seg;
                        it.term_begin = g.term_begin;
                        it.term_count = g.term_count;
                        it.tile_begin = tb;
                        it.tile_end = std::min(nt, tb + per);
                        it.out_slot = S.n_rows++;
                        S.item_cost.push_back(hg.cost * (it.tile_end - it.tile_begin) / nt + 1);
                        S.items.push_back(it);
                    }
                }
            }
            dq[q].part_count = S.n_rows - dq[q].part_begin;
            if (dq[q].part_count != 1) S.direct = false;
        }
    });
    uint32_t n_rows = 0, n_witems = 0, n_items = 0, n_bgroups = 0;
    bool direct = n_queries > 0;
    for (unsigned s = 0; s < width; s++) {
        PrepSlice& S = P.slices[s];
        S.row_off = n_rows; S.item_off = n_items; S.bgroup_off = n_bgroups;
        n_rows += S.n_rows; n_witems += (uint32_t)S.witems.size(); n_items += (uint32_t)S.items.size(); n_bgroups += (uint32_t)S.bgroups.size();
        direct = direct && S.direct;
    }
    // launch order of the wave items: narrow (<= 16 terms) before wide, longest estimated run time first, ties in query order
    uint32_t n_class[3] = {0, 0, 0};
    {
        uint32_t pos = 0;
        P.bucket_pos.assign(kOrderBuckets + 1, 0u);
        for (uint32_t half = 0; half < 2; half++) {
            for (uint32_t bkt = 0; bkt < kOrderBuckets; bkt++) {
                if (half == 0) P.bucket_pos[bkt] = pos;
                for (unsigned s = 0; s < width; s++) {
                    P.slices[s].start[half * kOrderBuckets + bkt] = pos;
                    pos += P.slices[s].hist[half * kOrderBuckets + bkt];
                }
            }
            if (half == 0) { n_class[0] = pos; P.bucket_pos[kOrderBuckets] = pos; }
        }
        n_class[1] = pos - n_class[0];
        // every wave item has exactly one launch position (the kernels trust the item array: an item lost or doubled here
        // would be a wild descriptor on the device)
        if (pos != n_witems) return fail(ctx, NS_E_STATE, "internal: launch order holds %u of %u work items", pos, n_witems);
        if (!auto_mode) { n_class[0] = n_class[1] = 0; }
    }
    // the workgroup-kernel items (fallback path: few): longest first, serially
    std::vector<DevItem> sorted_items;
    if (n_items) {
        struct Cost { uint64_t c; uint32_t slice, idx; };
        std::vector<Cost> ic;
        ic.reserve(n_items);
        for (unsigned s = 0; s < width; s++)
            for (uint32_t i = 0; i < P.slices[s].items.size(); i++) ic.push_back({P.slices[s].item_cost[i], s, i});
        std::stable_sort(ic.begin(), ic.end(), [](const Cost& a, const Cost& b) { return a.c > b.c; });
        sorted_items.resize(n_items);
        for (uint32_t i = 0; i < n_items; i++) {
            DevItem it = P.slices[ic[i].slice].items[ic[i].idx];
            it.out_slot = direct ? it.query : it.out_slot + P.slices[ic[i].slice].row_off;
            sorted_items[i] = it;
        }
    }

    ns_batch* b = new ns_batch();
    b->ctx = ctx;
    b->st = ctx->stream;
    if (ctx->overlap && ctx->stream == ctx->own_stream && ctx->alt_stream) {   // (an externally owned stream is never second-guessed)
        ctx->flip = !ctx->flip;
        if (ctx->flip) b->st = ctx->alt_stream;
    }
    b->Q = n_queries; b->K = k; b->flags = flags;
    b->variant = ctx->variant; b->tile_docs = tile_docs; b->hb = vd.hb;
    b->n_items = n_items; b->n_witems = n_witems;
    b->n_bgroups = n_bgroups; b->n_terms = n_dterms;
    for (int c = 0; c < 3; c++) b->n_class[c] = n_class[c];
    b->n_parts = direct ? 0 : n_rows;
    b->postings = postings_total;
    b->direct = direct;
    b->imp = all_imp && postings_total > 0;
    if (shared) {
        b->shared = true;
        b->n_share = (uint32_t)share_build.size();
        b->share_postings = share_postings;
        ctx->live_shared++;
    }
    b->pk = (all_pk && postings_total > 0) ? ctx->use_packed : 0;
    for (unsigned s2 = 0; s2 < width; s2++) b->pruned = b->pruned || P.slices[s2].any_pruned;

    // queries cut into many partial rows: joined by k_merge_wide, one workgroup each
    std::vector<uint32_t> wide_q;
    if (!direct)
        for (uint32_t q = 0; q < n_queries; q++)
            if (merge_is_wide(dq[q].part_count, k)) wide_q.push_back(q);
    b->n_wide_q = (uint32_t)wide_q.size();

    // One device block per batch: [descriptors, uploaded in one copy][scratch][hits | nhits | found, fetched in one copy]
    hipError_t e = hipSuccess;
    auto chk = [&](hipError_t r) { if (e == hipSuccess) e = r; };
    size_t off = 0;
    auto place = [&](size_t bytes) { const size_t o = off; off = (off + std::max<size_t>(bytes, 1) + 255) & ~(size_t)255; return o; };
    const size_t Qn = std::max<uint32_t>(n_queries, 1), Pn = std::max<uint32_t>(b->n_parts, 1);
    const size_t o_items = place((size_t)n_items * sizeof(DevItem));
    const size_t o_witems = place((size_t)n_witems * sizeof(DevWItem));
    const size_t o_terms = place((size_t)n_dterms * sizeof(DevTerm));
    const size_t o_groups = place((size_t)n_bgroups * sizeof(DevGroup));
    const size_t o_queries = place(dq.size() * sizeof(dq[0]));
    const size_t o_segs = place(segs.size() * sizeof(segs[0]));
    const size_t o_wideq = place(wide_q.size() * 4);
    const size_t o_share = place(shared ? (share_build.size() + 1) * sizeof(DevShare) : 0);
    const size_t up_bytes = off;
    const size_t o_bounds = place(bounds_total * 4);
    size_t o_phits = 0, o_pnhits = 0, o_pfound = 0, o_heads = 0;
    if (!direct) {
        o_phits = place(Pn * k * sizeof(Hit));
        o_pnhits = place(Pn * 4);
        o_pfound = place(Pn * 8);
        o_heads = place(Pn * 4);
    }
    const size_t o_hits = place(Qn * k * sizeof(Hit));
    const size_t o_nhits = place(Qn * 4);
    const size_t o_found = place(Qn * 8);
    char* base = nullptr;
    chk(batch_alloc(b, (void**)&base, off));
    if (e == hipSuccess) {
        b->d_items = (decltype(b->d_items))(base + o_items);
        b->d_witems = (decltype(b->d_witems))(base + o_witems);
        b->d_terms = (decltype(b->d_terms))(base + o_terms);
        b->d_groups = (decltype(b->d_groups))(base + o_groups);
        b->d_queries = (decltype(b->d_queries))(base + o_queries);
        b->d_segs = (decltype(b->d_segs))(base + o_segs);
        b->d_wide_q = (uint32_t*)(base + o_wideq);
        b->d_share = (DevShare*)(base + o_share);
        b->d_bounds = (uint32_t*)(base + o_bounds);
        if (!direct) {
            b->d_part_hits = (Hit*)(base + o_phits);
            b->d_part_nhits = (uint32_t*)(base + o_pnhits);
            b->d_part_found = (uint64_t*)(base + o_pfound);
            b->d_heads = (uint32_t*)(base + o_heads);
        }
        b->d_hits = (Hit*)(base + o_hits);
        b->d_nhits = (uint32_t*)(base + o_nhits);
        b->d_found = (uint64_t*)(base + o_found);
        b->out_span = o_found + Qn * 8 - o_hits;
        b->off_nhits = o_nhits - o_hits;
        b->off_found = o_found - o_hits;
        if (b->out_span <= kHostResultBytes && !ctx->down_owner) {
            if (ctx->h_down_cap < b->out_span) {
                if (ctx->h_down) (void)hipHostFree(ctx->h_down);
                ctx->h_down = nullptr; ctx->h_down_cap = 0;
                if (hipHostMalloc(&ctx->h_down, kHostResultBytes, hipHostMallocDefault) == hipSuccess) ctx->h_down_cap = kHostResultBytes;
                else { ctx->h_down = nullptr; (void)hipGetLastError(); }
            }
            if (ctx->h_down_cap >= b->out_span) {
                ctx->down_owner = b;
                b->d_hits = (Hit*)ctx->h_down;
                b->d_nhits = (uint32_t*)((char*)ctx->h_down + b->off_nhits);
                b->d_found = (uint64_t*)((char*)ctx->h_down + b->off_found);
            }
        }
    }
    if (e == hipSuccess && ctx->up_busy) {   // the previous batch's upload may still be reading the staging buffer
        chk(hipEventSynchronize(ctx->up_done));
        ctx->up_busy = false;
    }
    if (e == hipSuccess && up_bytes <= kStageMaxBytes && ctx->h_up_cap < up_bytes) {
        if (ctx->h_up) (void)hipHostFree(ctx->h_up);
        ctx->h_up = nullptr; ctx->h_up_cap = 0;
        const size_t cap = std::max<size_t>(up_bytes + up_bytes / 2, 1 << 16);
        if (hipHostMalloc(&ctx->h_up, cap, hipHostMallocDefault) == hipSuccess) ctx->h_up_cap = cap;
        else { ctx->h_up = nullptr; (void)hipGetLastError(); }
    }
    if (e == hipSuccess && !ctx->up_done) chk(hipEventCreateWithFlags(&ctx->up_done, hipEventDisableTiming));
    if (e == hipSuccess) {
        // ---- phase C: the descriptors go straight into the pinned staging buffer (or, for a batch too large for it, into
        // a host vector that is copied array by array), wave items at their place in the launch order ----
        const bool staged = ctx->h_up_cap >= up_bytes && ctx->up_done;
        std::vector<char> unstaged;
        if (!staged) unstaged.resize(up_bytes);
        char* hb = staged ? (char*)ctx->h_up : unstaged.data();
        // The dealing costs host time (a sort per class: +0.2 ms for cfg5's 16384 queries on 8 prepare threads).  A batch small
        // enough to be prepared by fewer than 4 threads over a cache-resident index gains ~1 % of kernel time from it and would
        // pay 0.3 ms of single-threaded sorting per 2048 queries — more than the batch's kernel — so it keeps the plain order
        // (2048-query batches pipelined: 0.71 ms per batch with the dealing, 0.44 without; profiles/r03/final_e2e_*.txt).
        uint64_t resident_bytes = 0;
        for (const ns_seg* sg_ : ctx->segs) if (sg_) resident_bytes += sg_->n_postings * 12ull;
        const bool deal = ctx->order_mode >= 1 && auto_mode && n_class[0] >= 64 && (width >= 4 || resident_bytes > (256ull << 20) || ctx->order_mode >= 2);
        if (deal && P.share_at.size() < n_witems) P.share_at.resize(n_witems);
        fork([&](unsigned si) {
            PrepSlice& S = P.slices[si];
            DevWItem* wdst = (DevWItem*)(hb + o_witems);
            for (size_t i = 0; i < S.witems.size(); i++) {
                DevWItem it = S.witems[i];
                it.out_slot = direct ? it.query : it.out_slot + S.row_off;
                const uint32_t bk = S.wbucket[i];
                const uint32_t at = S.start[((bk & 0x8000u) ? kOrderBuckets : 0) + (bk & 0x7FFFu)]++;
                wdst[at] = it;
                if (deal) P.share_at[at] = S.wshare[i];
            }
            if (!S.dterms.empty()) std::memcpy(hb + o_terms + (size_t)S.term_off * sizeof(DevTerm), S.dterms.data(), S.dterms.size() * sizeof(DevTerm));
            if (!S.bgroups.empty()) std::memcpy(hb + o_groups + (size_t)S.bgroup_off * sizeof(DevGroup), S.bgroups.data(), S.bgroups.size() * sizeof(DevGroup));
            for (uint32_t q = S.q0; q < S.q1; q++) dq[q].part_begin += S.row_off;
            if (S.q1 > S.q0) std::memcpy(hb + o_queries + (size_t)S.q0 * sizeof(DevQuery), dq.data() + S.q0, (size_t)(S.q1 - S.q0) * sizeof(DevQuery));
        });
        // ---- XCD dealing.  The launch order is longest-estimated-run-time first (2048 fine buckets).  Inside a coarse class of
        // 8 fine buckets (run times within ~19 % of each other) the order is free, and it is used for locality: workgroup i
        // runs on XCD i % 8, each XCD has its own 4 MB L2, and items that read the same bytes — same segment, same doc range
        // of the grid, same largest list: the shards of a hot list that dozens of queries of a batch share — should meet
        // in ONE L2 at about the same time, so that one of them pulls a line from HBM and the others hit it.  The items of a
        // class are sorted by their locality key (segment, then doc range, then a hash of the largest list), the sorted
        // sequence is cut into eight equal parts, and XCD x — the launch positions p with p % 8 == x — takes part x in
        // order: one L2 per part of the doc space, neighbours in time share lists.
        // Measured (profiles/r03): 20 x 1M-doc index, L2-miss traffic of the cfg5 launch 44.2 -> 31 GB, 7.05 -> 6.70 ms; the
        // 1M-doc index 2.61 -> 2.56 ms.  (A key quantised to eighths of the doc space lost 3 % there: the exact range matters.)
        // The classes are spread over the prepare threads; a class of n items costs one sort of n 64-bit words.
        if (deal) {
            DevWItem* wd = (DevWItem*)(hb + o_witems);
            // classes of 8 fine buckets while the index fits the 256 MiB Infinity Cache (an L2 miss is cheap there and the
            // longest-first order matters more), of 16 when it does not (20 x 1M docs: L2-miss traffic 31.6 -> 28.6 GB at
            // the same launch time; the 1M-doc index loses 3 % with 32, profiles/r03)
            const uint32_t shift = ctx->order_coarse_forced ? (uint32_t)ctx->order_coarse : (resident_bytes > (256ull << 20) ? 4u : 3u);
            const uint32_t n_cls = kOrderBuckets >> shift;
            if (P.deal_tmp.size() < width) { P.deal_tmp.resize(width); P.deal_key.resize(width); P.deal_alt.resize(width); P.deal_bins.resize(width); }
            fork([&](unsigned si) {
                std::vector<DevWItem>& tmp = P.deal_tmp[si];
                std::vector<uint64_t>& ord = P.deal_key[si];   // (key << 32 | index in the class): sorted = stable by key
                for (uint32_t c = si; c < n_cls; c += width) {
                    const uint32_t p0 = P.bucket_pos[c << shift], p1 = P.bucket_pos[(c + 1) << shift];
                    const uint32_t n = p1 - p0;
                    if (n < 16) continue;
                    ord.resize(n);
                    for (uint32_t i = 0; i < n; i++) ord[i] = ((uint64_t)P.share_at[p0 + i] << 32) | i;
                    if (n <= 4096) {
                        std::sort(ord.begin(), ord.end());
                    } else {
                        // a large class (all thin items of a batch have about the same run time: 17 000 items in one class of
                        // cfg5) would keep ONE prepare thread in a comparison sort for ~1 ms: two stable counting passes over
                        // the key's halves instead (the index in the low word is ascending already)
                        std::vector<uint64_t>& alt = P.deal_alt[si];
                        std::vector<uint32_t>& bins = P.deal_bins[si];
                        alt.resize(n);
                        bins.resize(65537);
                        for (int pass = 0; pass < 2; pass++) {
                            const int sh = 32 + 16 * pass;
                            std::fill(bins.begin(), bins.end(), 0u);
                            const uint64_t* src = pass ? alt.data() : ord.data();
                            uint64_t* dst = pass ? ord.data() : alt.data();
                            for (uint32_t i = 0; i < n; i++) bins[((src[i] >> sh) & 0xFFFFu) + 1u]++;
                            for (uint32_t b2 = 0; b2 < 65536; b2++) bins[b2 + 1] += bins[b2];
                            for (uint32_t i = 0; i < n; i++) dst[bins[(src[i] >> sh) & 0xFFFFu]++] = src[i];
                        }
                    }
                    tmp.assign(wd + p0, wd + p1);
                    uint32_t cur[8], end[8];
                    for (uint32_t x = 0; x < 8; x++) { cur[x] = (uint32_t)((uint64_t)n * x / 8); end[x] = (uint32_t)((uint64_t)n * (x + 1) / 8); }
                    for (uint32_t p = 0; p < n; p++) {
                        uint32_t x = (p0 + p) & 7u;
                        for (uint32_t tr = 0; tr < 8 && cur[x] >= end[x]; tr++) x = (x + 1) & 7u;   // a part one item short of its slots
                        wd[p0 + p] = tmp[(uint32_t)ord[cur[x]++]];
                    }
                }
            });
        }
        if (n_items) std::memcpy(hb + o_items, sorted_items.data(), (size_t)n_items * sizeof(DevItem));
        if (!segs.empty()) std::memcpy(hb + o_segs, segs.data(), segs.size() * sizeof(segs[0]));
        if (!wide_q.empty()) std::memcpy(hb + o_wideq, wide_q.data(), wide_q.size() * 4);
        if (shared) {
            std::memcpy(hb + o_share, share_build.data(), share_build.size() * sizeof(DevShare));
            const DevShare sentinel{0u, 0u, 0.0f, 0u, (uint32_t)share_postings};
            std::memcpy(hb + o_share + share_build.size() * sizeof(DevShare), &sentinel, sizeof(DevShare));
        }
        if (staged) {
            // a small upload is pulled by a kernel (the pinned buffer is device-addressable): a DMA-engine copy
            // followed by a kernel costs ~11 us of cross-engine hand-over, more than the copy itself
            // ... and with batches alternating between two streams (ns_ctx_set_overlap) EVERY upload is pulled: copies of all
            // streams go through one in-order DMA queue, where batch i+1's upload would sit behind batch i's result copy
            // — i.e. wait for batch i's kernels — and batch i+1's kernels with it (measured: 0.12 ms between consecutive
            // batches on two streams, as much as on one)
            // A LARGE upload (a 16384-query batch: 2.7 MB; 4096 queries over 8 segments: 6.5 MB = 0.2-0.25 ms) is pulled on the
            // ctx's pull stream and the batch's stream waits for it: it then runs next to the previous batch's scoring kernel
            // on one stream as on two.
            if (up_bytes > kPullUploadBytes && ctx->pull_stream) {
                hipLaunchKernelGGL(k_pull, dim3((uint32_t)((up_bytes / 16 + 255) / 256)), dim3(256), 0, ctx->pull_stream,
                                   (uint4*)base, (const uint4*)hb, (uint32_t)(up_bytes / 16));
                chk(hipEventRecord(ctx->up_done, ctx->pull_stream));
                chk(hipStreamWaitEvent(b->st, ctx->up_done, 0));
            } else {
                if (up_bytes <= kPullUploadBytes || ctx->overlap)
                    hipLaunchKernelGGL(k_pull, dim3((uint32_t)((up_bytes / 16 + 255) / 256)), dim3(256), 0, b->st,
                                       (uint4*)base, (const uint4*)hb, (uint32_t)(up_bytes / 16));
                else
                    chk(hipMemcpyAsync(base, hb, up_bytes, hipMemcpyHostToDevice, b->st));
                chk(hipEventRecord(ctx->up_done, b->st));
            }
            if (e == hipSuccess) ctx->up_busy = true;
        } else {
            chk(hipMemcpyAsync(base, hb, up_bytes, hipMemcpyHostToDevice, b->st));
            chk(hipStreamSynchronize(b->st));   // the copy reads a host vector that dies with this call
        }
    }
    if (e != hipSuccess) {
        int rc = fail(ctx, e == hipErrorOutOfMemory ? NS_E_NOMEM : NS_E_HIP, "ns_batch_prepare: %s", hipGetErrorString(e));
        ns_batch_destroy(b);
        return rc;
    }
    b->o_hits = b->d_hits; b->o_nhits = b->d_nhits; b->o_found = b->d_found;
    *out = b;
    return NS_OK;
}

extern "C" int ns_batch_bind_outputs(ns_batch* b, void* d_hits, void* d_nhits, void* d_found) {
    if (!b) return NS_E_INVAL;
    b->o_hits = d_hits ? (Hit*)d_hits : b->d_hits;
    b->o_nhits = d_nhits ? (uint32_t*)d_nhits : b->d_nhits;
    b->o_found = d_found ? (uint64_t*)d_found : b->d_found;
    return NS_OK;
}

extern "C" int ns_batch_run(ns_batch* b, int run_flags) {
    if (!b) return NS_E_INVAL;
    ns_ctx* ctx = b->ctx;
    const int timed = run_flags & NS_RUN_TIMED;
    // preconditions are checked BEFORE anything is enqueued, and the previous run's completion event stops counting from
    // here on: whatever early return follows, ns_batch_destroy then falls back to synchronising the batch's stream instead
    // of trusting an event that lies before kernels of this run
    if (run_flags & NS_RUN_FETCH) {
        const bool own_outputs = b->o_hits == b->d_hits && b->o_nhits == b->d_nhits && b->o_found == b->d_found;
        if (!own_outputs) return fail(ctx, NS_E_STATE, "NS_RUN_FETCH: the batch writes into caller-bound device buffers (ns_batch_bind_outputs); there is nothing to fetch");
    }
    b->done_recorded = false;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = b->st;
    const bool and_mode = (b->flags & NS_FLAG_AND) != 0;
    hipEvent_t* ev = nullptr;
    if (timed) {
        while (b->ev_pool.size() < b->ev_pending + 4) {
            hipEvent_t e = nullptr;
            HIPCHK(ctx, hipEventCreate(&e));
            b->ev_pool.push_back(e);
        }
        ev = b->ev_pool.data() + b->ev_pending;
        b->ev_pending += 4;
    }
    if (timed) HIPCHK(ctx, hipEventRecord(ev[0], st));
    if (b->n_bgroups)
        hipLaunchKernelGGL(k_bounds, dim3(b->n_bgroups), dim3(128), 0, st, b->d_groups, b->d_terms, b->d_segs, b->d_bounds, b->tile_docs);
    if (timed) HIPCHK(ctx, hipEventRecord(ev[1], st));
    // shared term scores: every distinct list of the batch once, on every run, inside the scoring kernel's timed span
    if (b->shared && b->n_share)
        hipLaunchKernelGGL(k_share_scores, dim3((uint32_t)((b->share_postings + 1023) / 1024)), dim3(256), 0, st, b->d_share, b->n_share, b->d_segs);
    Hit* sh = b->direct ? b->o_hits : b->d_part_hits;
    uint32_t* sn = b->direct ? b->o_nhits : b->d_part_nhits;
    uint64_t* sf = b->direct ? b->o_found : b->d_part_found;
    if (b->n_witems && b->variant == 0) {
        // auto mode: ONE launch; each wave picks the body that suits its item (DevWItem::whole bit 1)
        dim3 grid((b->n_witems + 3) / 4), block(64 * kUscoreWavesPerBlock);
        // K <= 64: a 128-entry candidate buffer is enough (K + 64 appended per step at most) and its
        // smaller LDS footprint admits more workgroups per CU.  Groups of <= 16 terms (all but exotic
        // queries) run in the instantiation with 16-entry term tables; the rest in the 64-entry one.
#define NS_U2(CBV, TM, N, PTR, IMPV, PKV)                                                                         \
        {                                                                                                          \
            dim3 g_(((N) + kUscoreWavesPerBlock - 1) / kUscoreWavesPerBlock);                                     \
            if (and_mode) hipLaunchKernelGGL((k_uscore<512, 192, true, CBV, TM, IMPV, PKV>), g_, block, 0, st, (PTR), (N), b->d_terms, b->d_segs, sh, sn, sf, b->K); \
            else hipLaunchKernelGGL((k_uscore<512, 192, false, CBV, TM, IMPV, PKV>), g_, block, 0, st, (PTR), (N), b->d_terms, b->d_segs, sh, sn, sf, b->K);         \
        }
#define NS_U(CBV, TM, N, PTR)                                                                                      \
        {                                                                                                          \
            if (b->imp && b->pk) NS_U2(CBV, TM, N, PTR, true, 1)          /* scores come with the block: no norms at all */ \
            else if (b->imp) NS_U2(CBV, TM, N, PTR, true, 0)                                                       \
            else if (b->pk == 2) NS_U2(CBV, TM, N, PTR, false, 2)                                                  \
            else if (b->pk == 1) NS_U2(CBV, TM, N, PTR, false, 1)                                                  \
            else NS_U2(CBV, TM, N, PTR, false, 0)                                                                  \
        }
        const uint32_t n_narrow = b->n_class[0], n_wide = b->n_witems - b->n_class[0];
        (void)grid;
        if (b->K <= 32) {   // the buffer is shrunk to K when it holds more than CB - 64 entries: CB = 128 needs K well below 64
            if (n_narrow) NS_U(128, 16, n_narrow, b->d_witems);
            if (n_wide) NS_U(128, 64, n_wide, b->d_witems + n_narrow);
        } else {
            if (n_narrow) NS_U(256, 16, n_narrow, b->d_witems);
            if (n_wide) NS_U(256, 64, n_wide, b->d_witems + n_narrow);
        }
#undef NS_U
#undef NS_U2
    }
#ifdef NS_VARIANTS
    else if (b->n_witems) {
        const VariantDesc wv = kVariants[b->variant];
#define NS_D(HH, FF) launch_dscore<HH, FF>(and_mode, b->n_witems, st, b->d_witems, b->d_terms, b->d_segs, sh, sn, sf, b->K)
        if (wv.d == 1) {
            if (wv.hb == 512) launch_tscore<512>(and_mode, b->n_witems, st, b->d_witems, b->d_terms, b->d_segs, sh, sn, sf, b->K);
            else if (wv.hb == 2048) launch_tscore<2048>(and_mode, b->n_witems, st, b->d_witems, b->d_terms, b->d_segs, sh, sn, sf, b->K);
            else launch_tscore<1024>(and_mode, b->n_witems, st, b->d_witems, b->d_terms, b->d_segs, sh, sn, sf, b->K);
        } else if (wv.d == 0) {
            switch (b->variant) {
                case 13: NS_D(256, 64); break;
                case 14: NS_D(1024, 256); break;
                case 15: NS_D(512, 64); break;
                case 16: NS_D(512, 256); break;
                case 17: NS_D(1024, 128); break;
                default: NS_D(512, 128); break;
            }
        }
#undef NS_D
    }
#endif
    if (b->n_items) {   // term groups of more than 64 terms (and, in the variants build, every group of variants 1-4): the workgroup-tile kernel
#ifdef NS_VARIANTS
        const VariantDesc vd = kVariants[b->variant];
        if (vd.nt == 1024) launch_score<1024, 12, 4>(and_mode, b->n_items, st, b->d_items, b->d_terms, b->d_segs, b->d_bounds, sh, sn, sf, b->K);
        else if (vd.nt == 256) launch_score<256, 16, 4>(and_mode, b->n_items, st, b->d_items, b->d_terms, b->d_segs, b->d_bounds, sh, sn, sf, b->K);
        else if (vd.spt == 16) launch_score<512, 16, 8>(and_mode, b->n_items, st, b->d_items, b->d_terms, b->d_segs, b->d_bounds, sh, sn, sf, b->K);
        else
#endif
        launch_score<512, 12, 4>(and_mode, b->n_items, st, b->d_items, b->d_terms, b->d_segs, b->d_bounds, sh, sn, sf, b->K);
    }
    if (timed) HIPCHK(ctx, hipEventRecord(ev[2], st));
    if (!b->direct && b->Q > b->n_wide_q)
        hipLaunchKernelGGL(k_merge, dim3((b->Q + 3) / 4), dim3(256), 0, st, b->d_queries, b->Q, b->d_part_hits, b->d_part_nhits,
                           b->d_part_found, b->o_hits, b->o_nhits, b->o_found, b->K, b->d_heads);
    if (!b->direct && b->n_wide_q)
        hipLaunchKernelGGL(k_merge_wide, dim3(b->n_wide_q), dim3(256), 0, st, b->d_queries, b->d_wide_q, b->d_part_hits, b->d_part_nhits,
                           b->d_part_found, b->o_hits, b->o_nhits, b->o_found, b->K, b->d_heads);
    if (timed) HIPCHK(ctx, hipEventRecord(ev[3], st));
    HIPCHK(ctx, hipGetLastError());
    b->ran = true;
    if (run_flags & NS_RUN_FETCH) {
        if (b->Q && ctx->down_owner != b) {   // (a small batch that owns h_down already has its results in host memory)
            if (b->down_slot < 0) {
                int slot = -1;
                for (size_t i = 0; i < ctx->down_slots.size(); i++)
                    if (!ctx->down_slots[i].busy && ctx->down_slots[i].cap >= b->out_span) { slot = (int)i; break; }
                if (slot < 0) {
                    for (size_t i = 0; i < ctx->down_slots.size() && slot < 0; i++)
                        if (!ctx->down_slots[i].busy) {   // grow an idle slot
                            if (ctx->down_slots[i].p) (void)hipHostFree(ctx->down_slots[i].p);
                            ctx->down_slots[i] = ns_ctx::DownSlot{};
                            slot = (int)i;
                        }
                    if (slot < 0) {
                        if (ctx->down_slots.size() >= 8) return fail(ctx, NS_E_STATE, "NS_RUN_FETCH: more than 8 batches between run and fetch");
                        ctx->down_slots.emplace_back();
                        slot = (int)ctx->down_slots.size() - 1;
                    }
                    const size_t cap = std::max<size_t>(b->out_span + b->out_span / 4, 1 << 16);
                    if (hipHostMalloc(&ctx->down_slots[(size_t)slot].p, cap, hipHostMallocDefault) != hipSuccess) {
                        ctx->down_slots[(size_t)slot].p = nullptr;
                        (void)hipGetLastError();
                        return fail(ctx, NS_E_NOMEM, "NS_RUN_FETCH: pinned result buffer of %zu bytes", cap);
                    }
                    ctx->down_slots[(size_t)slot].cap = cap;
                }
                ctx->down_slots[(size_t)slot].busy = true;
                b->down_slot = slot;
            }
            HIPCHK(ctx, hipMemcpyAsync(ctx->down_slots[(size_t)b->down_slot].p, b->d_hits, b->out_span, hipMemcpyDeviceToHost, st));
        }
    }
    // completion event of THIS run: destroy (and a NS_RUN_FETCH fetch) wait for it instead of for the whole stream
    if (!b->done) HIPCHK(ctx, hipEventCreateWithFlags(&b->done, hipEventDisableTiming));
    HIPCHK(ctx, hipEventRecord(b->done, st));
    b->done_recorded = true;
    b->fetch_enqueued = (run_flags & NS_RUN_FETCH) != 0;
    return NS_OK;
}

extern "C" void* ns_batch_stream(ns_batch* b) { return b ? (void*)b->st : nullptr; }

// Diagnostic: device time between the END of `prev`'s last timed run (after its last kernel) and the START of `next`'s
// (before its first kernel) — the idle or overlapped time between two batches of a pipelined loop.  Both must have been
// run with NS_RUN_TIMED and must still exist; NS_E_STATE while `next` has not started yet.
extern "C" int ns_batch_gap_ms(ns_batch* prev, ns_batch* next, float* ms) {
    if (!prev || !next || !ms) return NS_E_INVAL;
    if (prev->ev_pool.size() < 4 || next->ev_pool.size() < 4) return fail(next->ctx, NS_E_STATE, "ns_batch_gap_ms: both batches need a timed run");
    const hipError_t e = hipEventElapsedTime(ms, prev->ev_pool[3], next->ev_pool[0]);
    if (e == hipErrorNotReady) { (void)hipGetLastError(); return NS_E_STATE; }
    if (e != hipSuccess) return fail(next->ctx, NS_E_HIP, "ns_batch_gap_ms: %s", hipGetErrorString(e));
    return NS_OK;
}


// reads the HIP-event timings of the runs since the last call (all of them lie before the point the caller has waited for)
static int batch_collect_timings(ns_batch* b) {
    ns_ctx* ctx = b->ctx;
    for (size_t i = 0; i + 4 <= b->ev_pending; i += 4) {
        HIPCHK(ctx, hipEventElapsedTime(&b->last_score_ms, b->ev_pool[i + 1], b->ev_pool[i + 2]));
        HIPCHK(ctx, hipEventElapsedTime(&b->last_total_ms, b->ev_pool[i], b->ev_pool[i + 3]));
        b->sum_score_ms += b->last_score_ms;
        b->sum_total_ms += b->last_total_ms;
        b->timed_runs++;
    }
    b->ev_pending = 0;
    return NS_OK;
}

extern "C" int ns_batch_sync(ns_batch* b) {
    if (!b) return NS_E_INVAL;
    ns_ctx* ctx = b->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(b->st));
    return batch_collect_timings(b);
}

extern "C" int ns_batch_fetch(ns_batch* b, ns_hit* hits_out, uint32_t* nhits_out, uint64_t* found_out) {
    if (!b) return NS_E_INVAL;
    ns_ctx* ctx = b->ctx;
    if (!b->ran) return fail(ctx, NS_E_STATE, "ns_batch_fetch before ns_batch_run");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = b->st;
    const bool own_outputs = b->o_hits == b->d_hits && b->o_nhits == b->d_nhits && b->o_found == b->d_found;
    if (b->fetch_enqueued && b->done_recorded && own_outputs) {   // NS_RUN_FETCH: wait for this batch alone; its results are in (or on their way to) pinned host memory
        HIPCHK(ctx, hipEventSynchronize(b->done));
        int rc = batch_collect_timings(b);
        if (rc != NS_OK) return rc;
        if (b->Q) {
            const char* h = (ctx->down_owner == b) ? (const char*)ctx->h_down : (const char*)ctx->down_slots[(size_t)b->down_slot].p;
            if (hits_out) std::memcpy(hits_out, h, (size_t)b->Q * b->K * sizeof(Hit));
            if (nhits_out) std::memcpy(nhits_out, h + b->off_nhits, (size_t)b->Q * 4);
            if (found_out) std::memcpy(found_out, h + b->off_found, (size_t)b->Q * 8);
        }
        // the pinned slot goes back to the ctx with the fetch (not only at destroy): a serving loop that fetches promptly may
        // keep any number of batches alive.  A second fetch of this run takes the ordinary path (the device buffers still
        // hold the results); the next NS_RUN_FETCH run acquires a slot again.
        if (b->down_slot >= 0) { ctx->down_slots[(size_t)b->down_slot].busy = false; b->down_slot = -1; }
        b->fetch_enqueued = false;
        return NS_OK;
    }
    if (b->Q && own_outputs && ctx->down_owner == b) {   // the results are already in host memory
        int rc = ns_batch_sync(b);
        if (rc != NS_OK) return rc;
        const char* h = (const char*)ctx->h_down;
        if (hits_out) std::memcpy(hits_out, h, (size_t)b->Q * b->K * sizeof(Hit));
        if (nhits_out) std::memcpy(nhits_out, h + b->off_nhits, (size_t)b->Q * 4);
        if (found_out) std::memcpy(found_out, h + b->off_found, (size_t)b->Q * 8);
        return NS_OK;
    }
    if (b->Q && own_outputs && !ctx->down_owner && b->out_span <= kStageMaxBytes) {
        if (ctx->h_down_cap < b->out_span) {
            if (ctx->h_down) (void)hipHostFree(ctx->h_down);
            ctx->h_down = nullptr; ctx->h_down_cap = 0;
            const size_t cap = std::max<size_t>(b->out_span + b->out_span / 2, 1 << 16);
            if (hipHostMalloc(&ctx->h_down, cap, hipHostMallocDefault) == hipSuccess) ctx->h_down_cap = cap;
            else { ctx->h_down = nullptr; (void)hipGetLastError(); }
        }
        if (ctx->h_down_cap >= b->out_span) {
            HIPCHK(ctx, hipMemcpyAsync(ctx->h_down, b->d_hits, b->out_span, hipMemcpyDeviceToHost, st));
            int rc = ns_batch_sync(b);
            if (rc != NS_OK) return rc;
            const char* h = (const char*)ctx->h_down;
            if (hits_out) std::memcpy(hits_out, h, (size_t)b->Q * b->K * sizeof(Hit));
            if (nhits_out) std::memcpy(nhits_out, h + b->off_nhits, (size_t)b->Q * 4);
            if (found_out) std::memcpy(found_out, h + b->off_found, (size_t)b->Q * 8);
            return NS_OK;
        }
    }
    if (b->Q) {
        if (hits_out) HIPCHK(ctx, hipMemcpyAsync(hits_out, b->o_hits, (size_t)b->Q * b->K * sizeof(Hit), hipMemcpyDeviceToHost, st));
        if (nhits_out) HIPCHK(ctx, hipMemcpyAsync(nhits_out, b->o_nhits, (size_t)b->Q * 4, hipMemcpyDeviceToHost, st));
        if (found_out) HIPCHK(ctx, hipMemcpyAsync(found_out, b->o_found, (size_t)b->Q * 8, hipMemcpyDeviceToHost, st));
    }
    return ns_batch_sync(b);
}

extern "C" int ns_batch_get_info(ns_batch* b, ns_batch_info* info) {
    if (!b || !info) return NS_E_INVAL;
    info->postings = b->postings;
    info->algo_bytes = b->postings * 8;
    info->n_queries = b->Q;
    info->n_items = b->n_items + b->n_witems;
    info->n_term_refs = b->n_terms;
    info->tile_docs = b->tile_docs;
    info->k = b->K;
    info->flags = b->flags | (b->imp ? NS_INFO_IMPACTS : 0u) | (b->pk ? NS_INFO_PACKED : 0u) | (b->pruned ? NS_INFO_PRUNED : 0u) |
                  (b->shared ? NS_INFO_SHARED : 0u);
    info->shared_lists = b->n_share;
    info->shared_postings = b->share_postings;
    info->last_score_kernel_ms = b->last_score_ms;
    info->last_total_ms = b->last_total_ms;
    info->timed_runs = b->timed_runs;
    info->sum_score_kernel_ms = b->sum_score_ms;
    info->sum_total_ms = b->sum_total_ms;
    return NS_OK;
}

extern "C" int ns_search_batch(ns_ctx* ctx, const ns_query_desc* queries, const ns_term_ref* terms, uint32_t n_queries,
                               uint32_t k, ns_hit* hits_out, uint32_t* nhits_out, uint64_t* found_out, uint32_t flags) {
    ns_batch* b = nullptr;
    int rc = ns_batch_prepare(ctx, queries, terms, n_queries, k, flags, &b);
    if (rc != NS_OK) return rc;
    rc = ns_batch_run(b, 0);
    if (rc == NS_OK) rc = ns_batch_fetch(b, hits_out, nhits_out, found_out);
    ns_batch_destroy(b);
    return rc;
}


// ------------------------------------------------------------------------------------------------
// f3: forward.bin -> inverted lists (csrc/ns_invert.hip)
// `adopt` != nullptr: the inverted lists also become the posting stream of that segment (an upload in progress whose announced
// payload is n_pairs postings) by a device-to-device copy; postings_out may then be NULL.
static int invert_run(ns_ctx* ctx, const uint32_t* doc_term_counts, uint32_t n_docs, const uint32_t* pairs,
                      uint64_t n_pairs, uint32_t n_terms, uint32_t* df_out, void* postings_out, uint64_t* kept_out,
                      float* device_ms_out, ns_seg* adopt) {
    if (!ctx) return fail(nullptr, NS_E_INVAL, "ns_invert_forward: ctx is NULL");
    if ((n_docs && !doc_term_counts) || (n_pairs && !pairs) || (n_terms && !df_out) || !kept_out) return fail(ctx, NS_E_INVAL, "ns_invert_forward: null argument");
    if (n_pairs >= (1ull << 32) - kIvTile) return fail(ctx, NS_E_INVAL, "ns_invert_forward: %llu pairs; this build indexes pairs with 32 bits (split the segment)", (unsigned long long)n_pairs);
    if (n_terms == 0xFFFFFFFFu) return fail(ctx, NS_E_INVAL, "ns_invert_forward: n_terms too large");
    *kept_out = 0;
    if (device_ms_out) *device_ms_out = 0.0f;
    std::vector<uint64_t> prefix((size_t)n_docs + 1, 0);
    for (uint32_t d = 0; d < n_docs; d++) prefix[d + 1] = prefix[d] + doc_term_counts[d];
    if (prefix[n_docs] != n_pairs) return fail(ctx, NS_E_INVAL, "ns_invert_forward: the per-document counts sum to %llu, not to n_pairs = %llu", (unsigned long long)prefix[n_docs], (unsigned long long)n_pairs);
    if (n_terms) std::memset(df_out, 0, (size_t)n_terms * 4);
    if (!n_pairs) return NS_OK;
    if (!postings_out && !adopt) return fail(ctx, NS_E_INVAL, "ns_invert_forward: postings_out is NULL");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const uint32_t n = (uint32_t)n_pairs;
    const uint32_t n_tiles = (n + kIvTile - 1) / kIvTile;
    // Digit plan, from n_terms alone (no device -> host sync decides it): the keys are the termIds below n_terms — a pair with
    // any other termId is dropped by the reference (src/lexicon.cpp:69-70) and leaves the sort in the first pass — so
    // `bits` = the width of n_terms - 1, sorted in ceil(bits / 11) passes of 8 .. 11 bits, as even as possible.
    int bits = 1;
    while (bits < 32 && n_terms > 1 && ((n_terms - 1) >> bits) != 0) bits++;
    const int passes = std::max(1, (bits + 10) / 11);
    int pbits[3] = {8, 8, 8};
    {
        int left = bits;
        for (int p = 0; p < passes; p++) {
            const int share = (left + (passes - p) - 1) / (passes - p);
            pbits[p] = std::min(11, std::max(8, share));
            left = std::max(0, left - pbits[p]);
        }
    }
    size_t m_max = 0;
    for (int p = 0; p < passes; p++) m_max = std::max(m_max, ((size_t)1 << pbits[p]) * n_tiles);
    const uint32_t scan_blocks_max = (uint32_t)((m_max + 1023) / 1024);

    // one block from the ctx pool for all scratch arrays (a build loop inverts segment after segment of similar size;
    // ten hipMalloc + hipFree per call cost more than the device work)
    uint2 *d_pairs = nullptr, *d_vals[2] = {nullptr, nullptr};
    uint32_t *d_keys[2] = {nullptr, nullptr}, *d_df = nullptr, *d_first = nullptr, *d_hist = nullptr, *d_sums = nullptr, *d_kept = nullptr;
    uint64_t* d_prefix = nullptr;
    uint2* d_tile_docs = nullptr;
    // the documents that hold each tile's first and last pair (the host has the prefix sums; the first pass marks the
    // documents that start in between): the last d with prefix[d] <= i
    std::vector<uint32_t> tile_docs((size_t)n_tiles * 2);
    {
        uint32_t d = 0;
        auto doc_of = [&](uint64_t i) { while (d + 1 < n_docs && prefix[d + 1] <= i) d++; return d; };   // i ascends: one sweep over the documents
        for (uint32_t t = 0; t < n_tiles; t++) {
            const uint64_t i0 = (uint64_t)t * kIvTile, i1 = std::min<uint64_t>(i0 + kIvTile, n) - 1;
            tile_docs[2 * (size_t)t] = doc_of(i0);
            tile_docs[2 * (size_t)t + 1] = doc_of(i1);
        }
    }
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipError_t e = hipSuccess;
    auto chk = [&](hipError_t r) { if (e == hipSuccess) e = r; };
    size_t off = 0;
    auto place = [&](size_t bytes) { const size_t o = off; off = (off + std::max<size_t>(bytes, 1) + 255) & ~(size_t)255; return o; };
    const size_t nt1 = (size_t)std::max<uint32_t>(n_terms, 1);
    const size_t o_pairs = place((size_t)n * 8), o_v0 = place((size_t)n * 8), o_v1 = place(passes > 1 ? (size_t)n * 8 : 1);
    const size_t o_k0 = place((size_t)n * 4), o_k1 = place(passes > 1 ? (size_t)n * 4 : 1);
    const size_t o_df = place(nt1 * 4), o_first = place(nt1 * 4), o_hist = place(m_max * 4), o_sums = place((size_t)scan_blocks_max * 4), o_kept = place(4), o_prefix = place(prefix.size() * 8), o_tdocs = place((size_t)n_tiles * 8);
    const size_t block_bytes = off;
    char* blk = nullptr;
    chk(pool_alloc(ctx, (void**)&blk, block_bytes));
    if (e == hipSuccess) {
        d_pairs = (uint2*)(blk + o_pairs); d_vals[0] = (uint2*)(blk + o_v0); d_vals[1] = (uint2*)(blk + o_v1);
        d_keys[0] = (uint32_t*)(blk + o_k0); d_keys[1] = (uint32_t*)(blk + o_k1);
        d_df = (uint32_t*)(blk + o_df); d_first = (uint32_t*)(blk + o_first); d_hist = (uint32_t*)(blk + o_hist); d_sums = (uint32_t*)(blk + o_sums);
        d_kept = (uint32_t*)(blk + o_kept);
        d_prefix = (uint64_t*)(blk + o_prefix);
        d_tile_docs = (uint2*)(blk + o_tdocs);
    }
    chk(hipEventCreate(&ev0));
    chk(hipEventCreate(&ev1));
    uint2* d_final = nullptr;
    if (e == hipSuccess) {
        chk(hipMemcpyAsync(d_pairs, pairs, (size_t)n * 8, hipMemcpyHostToDevice, st));
        chk(hipMemcpyAsync(d_prefix, prefix.data(), prefix.size() * 8, hipMemcpyHostToDevice, st));
        chk(hipMemcpyAsync(d_tile_docs, tile_docs.data(), tile_docs.size() * 4, hipMemcpyHostToDevice, st));
        chk(hipMemsetAsync(d_df, 0, nt1 * 4, st));
        chk(hipMemsetAsync(d_first, 0xFF, nt1 * 4, st));
        chk(hipEventRecord(ev0, st));
        // pass p reads (p == 0: the pairs; else keys / vals buffer `cur`) and writes buffer `nxt`.  The number of items that
        // survive the first pass (the kept pairs) stays on the device: the first scan leaves it in d_kept, later kernels read it.
        int cur = -1;
        uint32_t shift = 0;
        for (int p = 0; p < passes; p++) {
            const bool first = p == 0, last = p == passes - 1;
            const int nxt = first ? 0 : (cur ^ 1);
            const size_t m = ((size_t)1 << pbits[p]) * n_tiles;
            const uint32_t scan_blocks = (uint32_t)((m + 1023) / 1024);
            const uint32_t* kin = first ? nullptr : d_keys[cur];
            const uint2* vin = first ? nullptr : d_vals[cur];
            const uint32_t* n_dev = first ? nullptr : d_kept;
#define NS_IV_HIST(B) hipLaunchKernelGGL((k_iv_hist_w<B>), dim3(n_tiles), dim3(256), 0, st, kin, first ? d_pairs : (const uint2*)nullptr, n, n_dev, n_terms, shift, d_hist, n_tiles)
#define NS_IV_PASS3(B, F, L) hipLaunchKernelGGL((k_iv_pass<B, F, L>), dim3(n_tiles), dim3(256), 0, st, d_pairs, d_prefix, d_tile_docs, n_terms, kin, vin, d_keys[nxt], d_vals[nxt], n, n_dev, shift, d_hist, n_tiles)
#define NS_IV_PASS(B) { if (first && last) NS_IV_PASS3(B, true, true); else if (first) NS_IV_PASS3(B, true, false); else if (last) NS_IV_PASS3(B, false, true); else NS_IV_PASS3(B, false, false); }
            switch (pbits[p]) {
                case 8: NS_IV_HIST(8); break;
                case 9: NS_IV_HIST(9); break;
                case 10: NS_IV_HIST(10); break;
                default: NS_IV_HIST(11); break;
            }
            hipLaunchKernelGGL(k_iv_scan_sums, dim3(scan_blocks), dim3(256), 0, st, d_hist, (uint32_t)m, d_sums);
            hipLaunchKernelGGL(k_iv_scan_top, dim3(1), dim3(1024), 0, st, d_sums, scan_blocks, first ? d_kept : (uint32_t*)nullptr);
            hipLaunchKernelGGL(k_iv_scan_apply, dim3(scan_blocks), dim3(256), 0, st, d_hist, (uint32_t)m, d_sums);
            switch (pbits[p]) {
                case 8: NS_IV_PASS(8); break;
                case 9: NS_IV_PASS(9); break;
                case 10: NS_IV_PASS(10); break;
                default: NS_IV_PASS(11); break;
            }
#undef NS_IV_HIST
#undef NS_IV_PASS
#undef NS_IV_PASS3
            shift += (uint32_t)pbits[p];
            cur = nxt;
        }
        d_final = d_vals[cur];
        // df from the sorted keys: a run's first pair records where it starts, its last pair where it ends (d_df holds the ends)
        if (n_terms) hipLaunchKernelGGL(k_iv_runs, dim3((n + 1023) / 1024), dim3(256), 0, st, d_keys[cur], n, n_terms, d_first, d_df, d_kept);
        chk(hipEventRecord(ev1, st));
        chk(hipGetLastError());
        std::vector<uint32_t> h_first(n_terms);
        if (n_terms) chk(hipMemcpyAsync(df_out, d_df, (size_t)n_terms * 4, hipMemcpyDeviceToHost, st));
        if (n_terms) chk(hipMemcpyAsync(h_first.data(), d_first, (size_t)n_terms * 4, hipMemcpyDeviceToHost, st));
        chk(hipStreamSynchronize(st));
        if (e == hipSuccess) {
            uint64_t kept = 0;
            for (uint32_t t = 0; t < n_terms; t++) df_out[t] = (h_first[t] == 0xFFFFFFFFu) ? 0u : df_out[t] - h_first[t] + 1u;
            for (uint32_t t = 0; t < n_terms; t++) kept += df_out[t];
            *kept_out = kept;   // the dropped pairs left the sort in the first pass
            if (kept && postings_out) chk(hipMemcpy(postings_out, d_final, (size_t)kept * 8, hipMemcpyDeviceToHost));
            if (adopt) {   // the lists stay on the device: they ARE the segment's posting stream
                if (kept) chk(hipMemcpyAsync(adopt->d_postings, d_final, (size_t)kept * 8, hipMemcpyDeviceToDevice, st));
                if (kept != adopt->n_postings) {   // dropped pairs: the stream is shorter than announced; move the padding
                    chk(hipMemsetAsync((char*)adopt->d_postings + kept * 8, 0xFF, kPadPostings * 8, st));
                    chk(hipMemsetAsync((char*)adopt->d_pnorm + kept * 4, 0, kPadPostings * 4, st));
                }
                chk(hipStreamSynchronize(st));
                if (e == hipSuccess) { adopt->n_postings = kept; adopt->filled = kept * 8; }
            }
            float ms = 0.0f;
            if (e == hipSuccess && hipEventElapsedTime(&ms, ev0, ev1) == hipSuccess && device_ms_out) *device_ms_out = ms;
        }
    }
    if (blk) { (void)hipStreamSynchronize(st); pool_free(ctx, blk, block_bytes); }   // nothing in flight uses the block any more
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
    if (e != hipSuccess) return fail(ctx, e == hipErrorOutOfMemory ? NS_E_NOMEM : NS_E_HIP, "ns_invert_forward: %s", hipGetErrorString(e));
    return NS_OK;
}

extern "C" int ns_invert_forward(ns_ctx* ctx, const uint32_t* doc_term_counts, uint32_t n_docs, const uint32_t* pairs,
                                 uint64_t n_pairs, uint32_t n_terms, uint32_t* df_out, void* postings_out, uint64_t* kept_out,
                                 float* device_ms_out) {
    return invert_run(ctx, doc_term_counts, n_docs, pairs, n_pairs, n_terms, df_out, postings_out, kept_out, device_ms_out, nullptr);
}

extern "C" int ns_segment_upload_inverted(ns_ctx* ctx, ns_seg* seg, const uint32_t* doc_term_counts, const uint32_t* pairs,
                                          uint64_t n_pairs, uint32_t n_terms, uint32_t* df_out, void* postings_out,
                                          uint64_t* kept_out, float* device_ms_out) {
    if (!ctx || !seg || seg->ctx != ctx || !seg->pending) return fail(ctx, NS_E_STATE, "ns_segment_upload_inverted: no upload in progress for this segment");
    if (seg->filled != 0) return fail(ctx, NS_E_STATE, "ns_segment_upload_inverted: the segment already received %llu payload bytes", (unsigned long long)seg->filled);
    if (seg->n_postings != n_pairs) return fail(ctx, NS_E_INVAL, "ns_segment_upload_inverted: %llu pairs, but ns_segment_upload_begin announced %llu postings", (unsigned long long)n_pairs, (unsigned long long)seg->n_postings);
    return invert_run(ctx, doc_term_counts, seg->n_docs, pairs, n_pairs, n_terms, df_out, postings_out, kept_out, device_ms_out, seg);
}

// ------------------------------------------------------------------------------------------------
// Segment-sharded multi-GPU: join the all-gathered per-rank rows (k_merge_ranks).  Device pointers; asynchronous on the ctx stream.
extern "C" int ns_merge_rank_rows(ns_ctx* ctx, const void* d_hits, const void* d_nhits, const void* d_found, uint32_t n_ranks,
                                  uint32_t n_queries, uint32_t k, const uint32_t* d_seg_map, uint32_t seg_map_stride,
                                  void* d_out_hits, void* d_out_nhits, void* d_out_found) {
    if (!ctx) return fail(nullptr, NS_E_INVAL, "ns_merge_rank_rows: ctx is NULL");
    if (n_ranks < 1 || n_ranks > 64) return fail(ctx, NS_E_INVAL, "ns_merge_rank_rows: %u ranks (1..64 supported)", n_ranks);
    if (k < 1 || k > NS_MAX_K) return fail(ctx, NS_E_INVAL, "k=%u outside [1,%u]", k, NS_MAX_K);
    if (!n_queries) return NS_OK;
    if (!d_hits || !d_nhits || !d_found || !d_out_hits || !d_out_nhits || !d_out_found) return fail(ctx, NS_E_INVAL, "ns_merge_rank_rows: null buffer");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_merge_ranks, dim3((n_queries + 3) / 4), dim3(256), 0, ctx->stream, (const Hit*)d_hits, (const uint32_t*)d_nhits,
                       (const uint64_t*)d_found, n_ranks, n_queries, k, d_seg_map, seg_map_stride, (Hit*)d_out_hits, (uint32_t*)d_out_nhits,
                       (uint64_t*)d_out_found);
    HIPCHK(ctx, hipGetLastError());
    return NS_OK;
}

// ------------------------------------------------------------------------------------------------
// f4: semantic expansion's similarity search (csrc/ns_sem.hip)
struct ns_sem {
    ns_ctx* ctx = nullptr;
    uint32_t rows = 0, dim = 0, rows_pad = 0;
    float* d_vt = nullptr;   // [dim][rows_pad]
};

extern "C" int ns_sem_upload(ns_ctx* ctx, const float* vecs, uint32_t n_rows, uint32_t dim, ns_sem** out) {
    if (!ctx) return fail(nullptr, NS_E_INVAL, "ns_sem_upload: ctx is NULL");
    if (!out || !vecs || !n_rows || !dim) return fail(ctx, NS_E_INVAL, "ns_sem_upload: empty table or null argument");
    *out = nullptr;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    ns_sem* s = new ns_sem();
    s->ctx = ctx; s->rows = n_rows; s->dim = dim; s->rows_pad = (n_rows + 63u) & ~63u;
    float* d_in = nullptr;
    hipError_t e = hipMalloc((void**)&s->d_vt, (size_t)dim * s->rows_pad * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&d_in, (size_t)n_rows * dim * 4);
    if (e == hipSuccess) e = hipMemcpyAsync(d_in, vecs, (size_t)n_rows * dim * 4, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_sem_transpose, dim3((s->rows_pad + 31) / 32, (dim + 31) / 32), dim3(256), 0, ctx->stream, d_in, s->d_vt, n_rows, dim, s->rows_pad);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_in);
    if (e != hipSuccess) {
        (void)hipFree(s->d_vt);
        delete s;
        return fail(ctx, e == hipErrorOutOfMemory ? NS_E_NOMEM : NS_E_HIP, "ns_sem_upload: %s", hipGetErrorString(e));
    }
    *out = s;
    return NS_OK;
}

extern "C" int ns_sem_release(ns_ctx* ctx, ns_sem* sem) {
    if (!ctx || !sem || sem->ctx != ctx) return fail(ctx, NS_E_INVAL, "ns_sem_release: table does not belong to this ctx");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    (void)hipFree(sem->d_vt);
    delete sem;
    return NS_OK;
}

extern "C" int ns_sem_topk(ns_ctx* ctx, ns_sem* sem, const float* qvecs, uint32_t n_q, uint32_t topk, float min_sim,
                           const uint32_t* ban_off, const uint32_t* ban_rows, uint32_t* rows_out, float* sims_out,
                           uint32_t* counts_out, float* device_ms_out) {
    if (!ctx || !sem || sem->ctx != ctx) return fail(ctx, NS_E_INVAL, "ns_sem_topk: table does not belong to this ctx");
    if (topk < 1 || topk > (uint32_t)kSemMaxK) return fail(ctx, NS_E_INVAL, "ns_sem_topk: topk=%u outside [1,%d]", topk, kSemMaxK);
    if (device_ms_out) *device_ms_out = 0.0f;
    if (!n_q) return NS_OK;
    if (!qvecs || !rows_out || !sims_out || !counts_out) return fail(ctx, NS_E_INVAL, "ns_sem_topk: null argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const uint32_t dim = sem->dim, rows = sem->rows, rp = sem->rows_pad;
    const uint32_t cap = ((rows + 63) / 64) * topk;   // keys a query's scan can leave: topk per wave of 64 rows
    const uint32_t n_groups = (n_q + kSemB - 1) / kSemB;
    const uint32_t n_ban = ban_off ? ban_off[n_q] : 0;
    // per-group ban offsets, rebased
    std::vector<uint32_t> goff((size_t)n_groups * (kSemB + 1), 0), grows(std::max<uint32_t>(n_ban, 1), 0);
    if (n_ban && !ban_rows) return fail(ctx, NS_E_INVAL, "ns_sem_topk: ban_rows is NULL");
    if (n_ban) std::memcpy(grows.data(), ban_rows, (size_t)n_ban * 4);
    for (uint32_t g = 0; g < n_groups; g++)
        for (uint32_t b = 0; b <= (uint32_t)kSemB; b++) {
            const uint32_t qi = std::min(g * kSemB + b, n_q);
            goff[(size_t)g * (kSemB + 1) + b] = ban_off ? ban_off[qi] : 0;
        }
    std::vector<float> qpad((size_t)n_groups * kSemB * dim, 0.0f);
    std::memcpy(qpad.data(), qvecs, (size_t)n_q * dim * 4);

    float *d_q = nullptr, *d_osims = nullptr;
    uint32_t *d_goff = nullptr, *d_grows = nullptr, *d_orows = nullptr, *d_ocnt = nullptr, *d_count = nullptr;
    uint64_t* d_cand = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipError_t e = hipSuccess;
    auto chk = [&](hipError_t r) { if (e == hipSuccess) e = r; };
    const size_t n_pad = (size_t)n_groups * kSemB;
    // one block from the ctx pool for all scratch arrays (every search of a serving loop expands its query)
    size_t off = 0;
    auto place = [&](size_t bytes) { const size_t o = off; off = (off + std::max<size_t>(bytes, 1) + 255) & ~(size_t)255; return o; };
    const size_t o_q = place(qpad.size() * 4), o_cand = place((size_t)kSemB * cap * 8), o_goff = place(goff.size() * 4), o_count = place((size_t)kSemB * 4),
                 o_grows = place(grows.size() * 4), o_orows = place(n_pad * topk * 4), o_osims = place(n_pad * topk * 4), o_ocnt = place(n_pad * 4);
    const size_t block_bytes = off;
    char* blk = nullptr;
    chk(pool_alloc(ctx, (void**)&blk, block_bytes));
    if (e == hipSuccess) {
        d_q = (float*)(blk + o_q); d_cand = (uint64_t*)(blk + o_cand); d_goff = (uint32_t*)(blk + o_goff); d_count = (uint32_t*)(blk + o_count);
        d_grows = (uint32_t*)(blk + o_grows); d_orows = (uint32_t*)(blk + o_orows); d_osims = (float*)(blk + o_osims); d_ocnt = (uint32_t*)(blk + o_ocnt);
    }
    chk(hipEventCreate(&ev0));
    chk(hipEventCreate(&ev1));
    if (e == hipSuccess) {
        chk(hipMemcpyAsync(d_q, qpad.data(), qpad.size() * 4, hipMemcpyHostToDevice, st));
        chk(hipMemcpyAsync(d_goff, goff.data(), goff.size() * 4, hipMemcpyHostToDevice, st));
        chk(hipMemcpyAsync(d_grows, grows.data(), grows.size() * 4, hipMemcpyHostToDevice, st));
        chk(hipMemsetAsync(d_count, 0, (size_t)kSemB * 4, st));   // k_sem_final_topk leaves it zero for the next group
        chk(hipEventRecord(ev0, st));
        for (uint32_t g = 0; g < n_groups; g++) {
            hipLaunchKernelGGL(k_sem_scan_topk, dim3((rows + 255) / 256), dim3(256), 0, st, sem->d_vt, rows, rp, dim, d_q + (size_t)g * kSemB * dim, min_sim,
                               d_goff + (size_t)g * (kSemB + 1), d_grows, topk, cap, d_cand, d_count);
            hipLaunchKernelGGL(k_sem_final_topk, dim3(kSemB), dim3(256), 0, st, d_cand, cap, d_count, topk, d_orows + (size_t)g * kSemB * topk, d_osims + (size_t)g * kSemB * topk, d_ocnt + (size_t)g * kSemB);
        }
        chk(hipEventRecord(ev1, st));
        chk(hipGetLastError());
        chk(hipMemcpyAsync(rows_out, d_orows, (size_t)n_q * topk * 4, hipMemcpyDeviceToHost, st));
        chk(hipMemcpyAsync(sims_out, d_osims, (size_t)n_q * topk * 4, hipMemcpyDeviceToHost, st));
        chk(hipMemcpyAsync(counts_out, d_ocnt, (size_t)n_q * 4, hipMemcpyDeviceToHost, st));
        chk(hipStreamSynchronize(st));
        float ms = 0.0f;
        if (e == hipSuccess && device_ms_out && hipEventElapsedTime(&ms, ev0, ev1) == hipSuccess) *device_ms_out = ms;
    }
    if (blk) { (void)hipStreamSynchronize(st); pool_free(ctx, blk, block_bytes); }
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
    if (e != hipSuccess) return fail(ctx, e == hipErrorOutOfMemory ? NS_E_NOMEM : NS_E_HIP, "ns_sem_topk: %s", hipGetErrorString(e));
    return NS_OK;
}

#ifdef NS_COUNT
// Diagnostic build only: the driver-stream body's event counters (ns_driver_kernel.hip), optionally reset.
extern "C" int ns_debug_counters(unsigned long long* out, int reset) {
    unsigned long long h[20];
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(ns::g_ns_cnt), sizeof(h)) != hipSuccess) return -1;
    if (out) std::memcpy(out, h, sizeof(h));
    if (reset) { std::memset(h, 0, sizeof(h)); if (hipMemcpyToSymbol(HIP_SYMBOL(ns::g_ns_cnt), h, sizeof(h)) != hipSuccess) return -1; }
    return 0;
}
extern "C" int ns_debug_tile_counters(unsigned long long* out, int reset) {
    unsigned long long h[12];
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(ns::g_ns_tcnt), sizeof(h)) != hipSuccess) return -1;
    if (out) std::memcpy(out, h, sizeof(h));
    if (reset) { std::memset(h, 0, sizeof(h)); if (hipMemcpyToSymbol(HIP_SYMBOL(ns::g_ns_tcnt), h, sizeof(h)) != hipSuccess) return -1; }
    return 0;
}
#endif
