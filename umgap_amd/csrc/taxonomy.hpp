// taxonomy.hpp — the taxonomy as it lives in HBM / Infinity Cache (gfx950 library internal).
//
// Replaces the start-up state of `umgap taxa2agg` (src/commands/taxa2agg.rs:104-109):
// TaxonList (src/taxon.rs:131-178), TaxonTree + snapping (src/taxon.rs:214-302) and the
// per-aggregator tables (parents: src/tree/lca.rs:25-30; ancestors: src/rmq/rtl.rs:27-31;
// Euler tour + RMQ: src/rmq/lca.rs:23-39) by ONE preorder-interval encoding:
//
//   by id   (len = max id + 1):  {pre, last}   pre = preorder index of the taxon (children in TSV
//                                 order, as the Euler tour visits them), last = largest preorder
//                                 index in its subtree; pre = NONE if the id is absent or not
//                                 reachable from the root
//   by pre  (n_reach entries):   parent_pre, depth, id, snap_valid, snap_ranked
//
//   a is an ancestor-or-self of b   <=>   pre[a] <= pre[b] <= last[a]
//   LCA(a, b), pre[a] < pre[b], a not an ancestor of b
//          = parent of the shallowest node with preorder index in (pre[a], pre[b]]
//            (every node in that range is a proper descendant of the LCA, and the range holds the
//             LCA's child on b's side).  Range-min over `depth` in O(1): the two partial 64-blocks
//            are read by the wave (coalesced), whole blocks in between come from a sparse table
//            of packed (depth << 32 | position) minima over 64-blocks (~5 MB, L2/Infinity Cache).
//
// ~22 B per taxon (2.5 M taxa: ~55 MB): resident in the 256 MiB Infinity Cache, top levels in L2.
#pragma once
#include <map>
#include <mutex>
#include <vector>

#include "common.hpp"

struct umgap_taxonomy {
    int device = 0;
    uint32_t root = 0, len = 0, n_reach = 0, n_rows = 0;
    // host copies (by id)
    std::vector<uint32_t> h_pre, h_last, h_depth, h_snap[2];
    std::vector<int8_t> h_seed_score;    // TaxonList::score by id for `seedextend -r` (12 under a named rank, -1 = None), id 0 always present
    int8_t* d_seed_score = nullptr;      // ... on the device, made at the first umgap_seedextend_ranked_dev
    // device, by id
    uint2* d_by_id = nullptr;            // {pre, last}
    uint8_t* d_err_kind = nullptr;       // 0 ok, 1 UnknownTaxon(payload), 2 panic (parent id out of bounds)
    uint32_t* d_err_payload = nullptr;   // first absent ancestor-or-self
    uint32_t* d_err_dist = nullptr;      // parent steps to reach it (BFS level in tree::Tree::new)
    // device, by pre
    uint32_t* d_parent_pre = nullptr;
    uint32_t* d_depth = nullptr;
    uint32_t* d_id_of_pre = nullptr;
    uint32_t* d_snap_of_pre[2] = {nullptr, nullptr};  // [0] valid, [1] valid && ranked
    uint32_t* d_psnap_of_pre[2] = {nullptr, nullptr}; // snap of the PARENT of the node at pre (LCA answer in one load)
    uint64_t* d_sparse = nullptr;                     // [levels][n_blocks] packed (depth << 32 | pos) block-range minima
    uint32_t n_blocks = 0, n_levels = 0;
    uint32_t* d_last_of_pre = nullptr;                // subtree end by pre
    uint64_t* d_sparse8 = nullptr;                    // [levels8][n_blocks8] packed (depth << 32 | pos) minima over 8-blocks: per-lane O(1) LCA
    uint32_t n_blocks8 = 0, n_levels8 = 0;
    hipStream_t stream = nullptr;
    // the upload lane and the compute lane of umgap_classify_records / umgap_aggregate (chunks c+1, c+2 upload while chunk c computes).
    // The upload lane carries COPIES ONLY: the runtime folds a process's streams onto four hardware compute queues, and a fill or an
    // event wait on a lane that shares its queue with the compute lane waits -- with every upload behind it -- for the running kernel
    // (tools/native/queue_probe.hip; seen inside bench.py: 193 instead of 360 M pairs/s)
    hipStream_t stream2 = nullptr, stream3 = nullptr;
    uint32_t* h_flags = nullptr;       // pinned: per-slot {bad offsets, unfit read} flags of umgap_classify_records
    void* h_stage = nullptr;           // pinned staging of umgap_classify_records for callers whose arrays are pageable (grow-only)
    size_t h_stage_bytes = 0;
    // staging for the host-pointer umgap_aggregate
    void* d_stage = nullptr;
    size_t d_stage_bytes = 0;
    // deferred-read lists of the general aggregation (aggregate_general.hip), ONE PER STREAM: launches on one stream are
    // ordered and may share a list, launches on different streams may overlap and must not (umgap::defer_list_for)
    struct DeferList { uint32_t* d = nullptr; size_t bytes = 0; };
    std::map<hipStream_t, DeferList> defer_lists;
    std::mutex defer_mutex;
    uint32_t* d_big_scratch = nullptr;   // global-memory working sets of the third aggregation pass (umgap::big_scratch_for), made on first use
    // indexes whose values are currently bound to this taxonomy (bind.hip): unbound again by umgap_taxonomy_free,
    // dropped from the list by umgap_index_free
    std::vector<umgap_index*> bound_indexes;
};

namespace umgap {
// The calling stream's deferred-read list, at least `bytes` long (grown with hipFree + hipMalloc, which wait for the
// device, so a launch still using the old list has finished).  nullptr + error set on failure.
uint32_t* defer_list_for(umgap_taxonomy* tx, hipStream_t stream, size_t bytes);
// Reads with more than 1024 distinct taxa (general aggregation only) are aggregated with their working set in global
// memory: BIG_SLOTS reads at a time, up to BIG_CAP distinct taxa each; beyond that UMGAP_READ_UNSUPPORTED.
constexpr uint32_t BIG_SLOTS = 16, BIG_CAP = 1u << 15, BIG_HCAP = 1u << 16;
uint32_t* big_scratch_for(umgap_taxonomy* tx);
// the upload lane and the compute lane of the host-pointer calls, each made under its own check: a handle whose second lane failed
// to come up must not run its kernels on the null stream next time
inline int ensure_lanes(umgap_taxonomy* tx) {
    for (hipStream_t* s : {&tx->stream2, &tx->stream3}) {
        if (*s) continue;
        const hipError_t e = hipStreamCreateWithFlags(s, hipStreamNonBlocking);
        if (e != hipSuccess) { *s = nullptr; return fail(UMGAP_ERR_DEVICE, std::string("hipStreamCreate: ") + hipGetErrorString(e)); }
    }
    return UMGAP_OK;
}

struct TaxView {
    const uint2* by_id;
    const uint8_t* err_kind;
    const uint32_t* err_payload;
    const uint32_t* err_dist;
    const uint32_t* parent_pre;
    const uint32_t* depth;
    const uint32_t* id_of_pre;
    const uint32_t* snap_of_pre;   // chosen by ranked_only
    const uint32_t* psnap_of_pre;  // chosen by ranked_only
    const uint64_t* sparse;
    const uint32_t* last_of_pre;
    const uint64_t* sparse8;
    uint32_t len, n_reach, n_blocks, n_blocks8;
};
inline TaxView view_of(const umgap_taxonomy* tx, int ranked_only) {
    return TaxView{tx->d_by_id, tx->d_err_kind, tx->d_err_payload, tx->d_err_dist, tx->d_parent_pre, tx->d_depth, tx->d_id_of_pre,
                   tx->d_snap_of_pre[ranked_only ? 1 : 0], tx->d_psnap_of_pre[ranked_only ? 1 : 0], tx->d_sparse,
                   tx->d_last_of_pre, tx->d_sparse8, tx->len, tx->n_reach, tx->n_blocks, tx->n_blocks8};
}
}  // namespace umgap
