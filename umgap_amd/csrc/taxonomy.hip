// taxonomy.hip — umgap_taxonomy_create: host-side derivation of the preorder-interval tables
// (taxonomy.hpp) and their upload.  Start-up cost only (src/commands/taxa2agg.rs:104-109).
#include <algorithm>

#include "taxonomy.hpp"

using namespace umgap;

namespace {
constexpr uint32_t NONE = UMGAP_NONE;

template <class T>
int upload(T** dst, const std::vector<T>& src) {
    size_t bytes = std::max<size_t>(src.size(), 1) * sizeof(T);
    hipError_t e = hipMalloc((void**)dst, bytes);
    if (e != hipSuccess) return fail(UMGAP_ERR_NOMEM, std::string("hipMalloc(taxonomy): ") + hipGetErrorString(e));
    if (!src.empty()) {
        e = hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice);
        if (e != hipSuccess) return fail(UMGAP_ERR_DEVICE, std::string("hipMemcpy(taxonomy): ") + hipGetErrorString(e));
    }
    return UMGAP_OK;
}
}  // namespace

extern "C" {

int umgap_taxonomy_create(const uint32_t* id, const uint32_t* parent, const uint8_t* valid, const uint8_t* rank, size_t n,
                          int device, umgap_taxonomy** out) {
    if (!out || (n && (!id || !parent || !valid || !rank))) return fail(UMGAP_ERR_INVALID, "umgap_taxonomy_create: null argument");
    // TaxonTree::new indexes taxons[0] (src/taxon.rs:226): empty input panics
    if (n == 0) return fail(UMGAP_ERR_PANIC, "index out of bounds: the len is 0 but the index is 0");
    uint32_t max_id = 0;
    for (size_t i = 0; i < n; i++) {
        if (id[i] == NONE || parent[i] == NONE) return fail(UMGAP_ERR_FORMAT, "taxon id >= 2^32-1 is not supported");
        max_id = std::max(max_id, id[i]);
    }
    const uint32_t len = max_id + 1;  // TaxonList::new (src/taxon.rs:135-145)
    // A repeated id.  The reference keeps the LAST row's name / rank / valid (TaxonList::new, src/taxon.rs:135-145) and BOTH child
    // links (TaxonTree::new, :228-241).  With the same parent in every copy that is the tree without the repeats: the second visit
    // of the subtree changes neither `ancestors` nor any first occurrence, and a range minimum that spans it passes through the
    // parent anyway -- so the link of the first copy is kept (its place among the siblings) with the attributes of the last.
    // Copies with DIFFERENT parents make a graph no preorder interval can hold: refused.
    std::vector<uint32_t> row_of_id(len, NONE);
    std::vector<uint8_t> repeat(n, 0);           // this row repeats an earlier one: no link of its own
    for (size_t i = 0; i < n; i++) {
        const uint32_t before = row_of_id[id[i]];
        if (before != NONE) {
            if (parent[before] != parent[i])
                return fail(UMGAP_ERR_FORMAT, "duplicate taxon id " + std::to_string(id[i]) + " with two different parents (refused; the reference would keep both child links)");
            repeat[i] = 1;
        }
        row_of_id[id[i]] = (uint32_t)i;
    }
    // root = the only taxon with id == parent (src/taxon.rs:228-246)
    uint32_t root = NONE;
    size_t n_roots = 0;
    for (size_t i = 0; i < n; i++)
        if (id[i] == parent[i] && !repeat[i]) { n_roots++; root = id[i]; }
    if (n_roots > 1) return fail(UMGAP_ERR_PANIC, "More than one root!");
    if (n_roots == 0) return fail(UMGAP_ERR_PANIC, "There's no root!");

    // children in row order, CSR keyed by parent id (parents >= len can never be reached)
    std::vector<uint32_t> child_start(len + 1, 0), child(n);
    for (size_t i = 0; i < n; i++)
        if (id[i] != parent[i] && parent[i] < len && !repeat[i]) child_start[parent[i] + 1]++;
    for (uint32_t p = 0; p < len; p++) child_start[p + 1] += child_start[p];
    {
        std::vector<uint32_t> cur(child_start.begin(), child_start.end() - 1);
        for (size_t i = 0; i < n; i++)
            if (id[i] != parent[i] && parent[i] < len && !repeat[i]) child[cur[parent[i]]++] = id[i];
    }

    auto* tx = new umgap_taxonomy();
    tx->device = device;
    tx->root = root;
    tx->len = len;
    tx->n_rows = (uint32_t)n;
    {   // TaxonList::score over TaxonList::new_with_unknown(taxa, true) (src/commands/seedextend.rs:77-83): the first ancestor-or-self
        // that is its own parent or carries a rank decides -- 12 under a named rank, None otherwise; an absent taxon on the way: None
        std::vector<int8_t>& sc = tx->h_seed_score;
        sc.assign(len, (int8_t)-2);                       // -2 = not computed
        std::vector<uint32_t> path;
        auto present = [&](uint32_t t) { return t < len && (row_of_id[t] != NONE || t == 0); };
        for (uint32_t t0 = 0; t0 < len; t0++) {
            if (sc[t0] != -2) continue;
            path.clear();
            int8_t res = -1;
            uint32_t cur = t0;
            for (size_t steps = 0; steps <= (size_t)len; steps++) {
                if (!present(cur)) { res = -1; break; }
                if (sc[cur] != -2) { res = sc[cur]; break; }
                const uint32_t row = row_of_id[cur];
                const uint32_t par = row == NONE ? 0u : parent[row];          // the added "unknown" taxon 0: its own parent, no rank
                const uint8_t rk = row == NONE ? (uint8_t)0 : rank[row];
                if (par == cur || rk != 0) { res = (rk != 0 && rk < 27) ? (int8_t)12 : (int8_t)-1; sc[cur] = res; break; }
                path.push_back(cur);
                cur = par;
            }
            for (uint32_t v : path) sc[v] = res;
        }
    }
    tx->h_pre.assign(len, NONE);
    tx->h_last.assign(len, NONE);
    tx->h_depth.assign(len, NONE);
    tx->h_snap[0].assign(len, NONE);
    tx->h_snap[1].assign(len, NONE);
    std::vector<uint32_t> parent_pre, depth_of_pre, id_of_pre, snap_pre[2];
    parent_pre.reserve(n); depth_of_pre.reserve(n); id_of_pre.reserve(n);
    snap_pre[0].reserve(n); snap_pre[1].reserve(n);

    // preorder DFS from the root, children in row order (= EulerIterator order, src/taxon.rs:309-392;
    // = with_filtered order, src/taxon.rs:251-271).  snapping: nearest ancestor-or-self passing
    // the filter, seeded with Some(root) (src/taxon.rs:279,294-301).
    auto passes = [&](uint32_t t, int ranked) {
        uint32_t r = row_of_id[t];
        return r != NONE && valid[r] && (!ranked || rank[r] != 0);
    };
    struct Frame { uint32_t node, cursor, snap0, snap1; };
    std::vector<Frame> stack;
    uint32_t counter = 0;
    auto enter = [&](uint32_t node, uint32_t inh0, uint32_t inh1, uint32_t par_pre, uint32_t depth) {
        uint32_t s0 = passes(node, 0) ? node : inh0, s1 = passes(node, 1) ? node : inh1;
        tx->h_pre[node] = counter;
        tx->h_depth[node] = depth;
        tx->h_snap[0][node] = s0;
        tx->h_snap[1][node] = s1;
        parent_pre.push_back(par_pre);
        depth_of_pre.push_back(depth);
        id_of_pre.push_back(node);
        snap_pre[0].push_back(s0);
        snap_pre[1].push_back(s1);
        counter++;
        stack.push_back({node, child_start[node], s0, s1});
    };
    enter(root, root, root, 0, 0);
    while (!stack.empty()) {
        Frame& f = stack.back();
        if (f.cursor < child_start[f.node + 1]) {
            uint32_t c = child[f.cursor++];
            if (tx->h_pre[c] != NONE) {  // cannot happen without duplicate ids; guard against cycles
                delete tx;
                return fail(UMGAP_ERR_FORMAT, "taxon " + std::to_string(c) + " is reachable twice");
            }
            uint32_t ppre = tx->h_pre[f.node], d = tx->h_depth[f.node] + 1, s0 = f.snap0, s1 = f.snap1;
            enter(c, s0, s1, ppre, d);
        } else {
            tx->h_last[f.node] = counter - 1;
            stack.pop_back();
        }
    }
    tx->n_reach = counter;

    // error tables for ids that are not reachable (tree::Tree::new src/tree/mod.rs:35-46 and
    // rmq::rtl src/rmq/rtl.rs:42-49 walk `parents`): first absent ancestor-or-self and the number of
    // parent steps to it; kind 2 when the chain leaves the list (index out of bounds panic).
    std::vector<uint8_t> err_kind(len, 0);
    std::vector<uint32_t> err_payload(len, 0), err_dist(len, 0);
    std::vector<uint8_t> state(len, 0);  // 0 unresolved, 1 on the current path, 2 resolved
    for (uint32_t t = 0; t < len; t++) {
        if (tx->h_pre[t] != NONE) { state[t] = 2; continue; }
        if (row_of_id[t] == NONE) { state[t] = 2; err_kind[t] = 1; err_payload[t] = t; err_dist[t] = 0; }
    }
    std::vector<uint32_t> path;
    for (uint32_t t0 = 0; t0 < len; t0++) {
        if (state[t0] == 2) continue;
        path.clear();
        uint32_t t = t0;
        uint8_t kind = 0;
        uint32_t payload = 0, dist = 0;
        for (;;) {
            if (t >= len) { kind = 2; payload = t; dist = 0; break; }  // parents[t] out of bounds
            if (state[t] == 2) { kind = err_kind[t]; payload = err_payload[t]; dist = err_dist[t]; break; }
            if (state[t] == 1) {
                delete tx;
                return fail(UMGAP_ERR_FORMAT, "parent cycle through taxon " + std::to_string(t) + " (the reference would not terminate)");
            }
            state[t] = 1;
            path.push_back(t);
            t = parent[row_of_id[t]];
        }
        // kind 0 here would mean the chain reaches a reachable node, i.e. t0 is reachable: impossible
        for (size_t k = path.size(); k-- > 0;) {
            dist++;
            uint32_t p = path[k];
            state[p] = 2;
            err_kind[p] = kind ? kind : 1;
            err_payload[p] = payload;
            err_dist[p] = dist;
        }
    }

    // parent-snap tables and the sparse table of block-range depth minima (taxonomy.hpp)
    std::vector<uint32_t> psnap_pre[2];
    for (int k = 0; k < 2; k++) {
        psnap_pre[k].resize(counter);
        for (uint32_t p = 0; p < counter; p++) psnap_pre[k][p] = snap_pre[k][parent_pre[p]];
    }
    const uint32_t n_blocks = (counter + 63) / 64;
    uint32_t n_levels = 1;
    while ((1u << n_levels) <= n_blocks) n_levels++;
    std::vector<uint64_t> sparse((size_t)n_levels * n_blocks, ~0ull);
    for (uint32_t b = 0; b < n_blocks; b++) {
        uint64_t best = ~0ull;
        for (uint32_t p = b * 64; p < counter && p < b * 64 + 64; p++) best = std::min(best, ((uint64_t)depth_of_pre[p] << 32) | p);
        sparse[b] = best;
    }
    for (uint32_t k = 1; k < n_levels; k++)
        for (uint32_t b = 0; b + (1u << k) <= n_blocks; b++)
            sparse[(size_t)k * n_blocks + b] = std::min(sparse[(size_t)(k - 1) * n_blocks + b], sparse[(size_t)(k - 1) * n_blocks + b + (1u << (k - 1))]);
    tx->n_blocks = n_blocks;
    tx->n_levels = n_levels;
    // subtree ends by pre, and a second sparse table over blocks of 8 for the per-lane LCA of the hybrid aggregators
    // (agg_general.hpp: two aligned 32-byte depth blocks + two table entries, all independent loads)
    std::vector<uint32_t> last_of_pre(counter);
    for (uint32_t p = 0; p < counter; p++) last_of_pre[p] = tx->h_last[id_of_pre[p]];
    const uint32_t n_blocks8 = (counter + 7) / 8;
    uint32_t n_levels8 = 1;
    while ((1u << n_levels8) <= n_blocks8) n_levels8++;
    std::vector<uint64_t> sparse8((size_t)n_levels8 * n_blocks8, ~0ull);
    for (uint32_t b8 = 0; b8 < n_blocks8; b8++) {
        uint64_t best = ~0ull;
        for (uint32_t p = b8 * 8; p < counter && p < b8 * 8 + 8; p++) best = std::min(best, ((uint64_t)depth_of_pre[p] << 32) | p);
        sparse8[b8] = best;
    }
    for (uint32_t k = 1; k < n_levels8; k++)
        for (uint32_t b8 = 0; b8 + (1u << k) <= n_blocks8; b8++)
            sparse8[(size_t)k * n_blocks8 + b8] = std::min(sparse8[(size_t)(k - 1) * n_blocks8 + b8], sparse8[(size_t)(k - 1) * n_blocks8 + b8 + (1u << (k - 1))]);
    tx->n_blocks8 = n_blocks8;
    tx->n_levels8 = n_levels8;
    depth_of_pre.resize((size_t)n_blocks8 * 8, NONE);   // whole 8-blocks on the device (the per-lane scan loads them as 2 x 16 B)

    int rc = use_device(device);
    if (rc) { delete tx; return rc; }
    std::vector<uint2> by_id(len);
    for (uint32_t t = 0; t < len; t++) by_id[t] = make_uint2(tx->h_pre[t], tx->h_last[t]);
    hipError_t e = hipStreamCreateWithFlags(&tx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete tx; return fail(UMGAP_ERR_DEVICE, std::string("hipStreamCreate: ") + hipGetErrorString(e)); }
    rc = upload(&tx->d_by_id, by_id);
    if (!rc) rc = upload(&tx->d_err_kind, err_kind);
    if (!rc) rc = upload(&tx->d_err_payload, err_payload);
    if (!rc) rc = upload(&tx->d_err_dist, err_dist);
    if (!rc) rc = upload(&tx->d_parent_pre, parent_pre);
    if (!rc) rc = upload(&tx->d_depth, depth_of_pre);
    if (!rc) rc = upload(&tx->d_id_of_pre, id_of_pre);
    if (!rc) rc = upload(&tx->d_snap_of_pre[0], snap_pre[0]);
    if (!rc) rc = upload(&tx->d_snap_of_pre[1], snap_pre[1]);
    if (!rc) rc = upload(&tx->d_psnap_of_pre[0], psnap_pre[0]);
    if (!rc) rc = upload(&tx->d_psnap_of_pre[1], psnap_pre[1]);
    if (!rc) rc = upload(&tx->d_sparse, sparse);
    if (!rc) rc = upload(&tx->d_last_of_pre, last_of_pre);
    if (!rc) rc = upload(&tx->d_sparse8, sparse8);
    if (rc) { umgap_taxonomy_free(tx); return rc; }
    *out = tx;
    return UMGAP_OK;
}

}  // extern "C"

namespace umgap {
uint32_t* defer_list_for(umgap_taxonomy* tx, hipStream_t stream, size_t bytes) {
    std::lock_guard<std::mutex> g(tx->defer_mutex);
    umgap_taxonomy::DeferList& L = tx->defer_lists[stream];
    if (bytes > L.bytes) {
        if (L.d) (void)hipFree(L.d);
        L.d = nullptr;
        L.bytes = 0;
        hipError_t e = hipMalloc((void**)&L.d, bytes);
        if (e != hipSuccess) { fail(UMGAP_ERR_NOMEM, std::string("hipMalloc(deferred-read list): ") + hipGetErrorString(e)); L.d = nullptr; return nullptr; }
        L.bytes = bytes;
    }
    return L.d;
}
uint32_t* big_scratch_for(umgap_taxonomy* tx) {
    std::lock_guard<std::mutex> g(tx->defer_mutex);
    if (!tx->d_big_scratch) {
        // 12 * cap + 2 * hcap words per slot (agg_general.hpp: ScratchGlobal::words)
        const size_t bytes = (size_t)BIG_SLOTS * (2 * (size_t)BIG_HCAP + 12 * (size_t)BIG_CAP) * sizeof(uint32_t);
        hipError_t e = hipMalloc((void**)&tx->d_big_scratch, bytes);
        if (e == hipSuccess) e = hipMemset(tx->d_big_scratch, 0, bytes);   // the hash tables start (and are handed back) clean
        if (e != hipSuccess) {
            if (tx->d_big_scratch) (void)hipFree(tx->d_big_scratch);
            tx->d_big_scratch = nullptr;
            fail(UMGAP_ERR_NOMEM, std::string("hipMalloc(global aggregation scratch): ") + hipGetErrorString(e));
        }
    }
    return tx->d_big_scratch;
}
}  // namespace umgap

extern "C" {

void umgap_taxonomy_free(umgap_taxonomy* tx) {
    if (!tx) return;
    (void)hipSetDevice(tx->device);
    // an index still bound to this taxonomy gets its taxon ids back first (its bound_to must not dangle)
    while (!tx->bound_indexes.empty()) {
        umgap_index* ix = tx->bound_indexes.back();
        if (umgap_index_unbind_taxonomy(ix, tx) != UMGAP_OK) { ix->bound_to = nullptr; tx->bound_indexes.pop_back(); }
    }
    (void)hipDeviceSynchronize();
    void* ptrs[] = {tx->d_by_id, tx->d_err_kind, tx->d_err_payload, tx->d_err_dist, tx->d_parent_pre, tx->d_depth,
                    tx->d_id_of_pre, tx->d_snap_of_pre[0], tx->d_snap_of_pre[1], tx->d_psnap_of_pre[0], tx->d_psnap_of_pre[1],
                    tx->d_sparse, tx->d_last_of_pre, tx->d_sparse8, tx->d_stage};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    for (auto& kv : tx->defer_lists)
        if (kv.second.d) (void)hipFree(kv.second.d);
    if (tx->d_big_scratch) (void)hipFree(tx->d_big_scratch);
    if (tx->d_seed_score) (void)hipFree(tx->d_seed_score);
    if (tx->stream) (void)hipStreamDestroy(tx->stream);
    if (tx->stream2) (void)hipStreamDestroy(tx->stream2);
    if (tx->stream3) (void)hipStreamDestroy(tx->stream3);
    if (tx->h_flags) (void)hipHostFree(tx->h_flags);
    if (tx->h_stage) (void)hipHostFree(tx->h_stage);
    delete tx;
}

uint32_t umgap_taxonomy_root(const umgap_taxonomy* tx) { return tx ? tx->root : NONE; }
uint32_t umgap_taxonomy_len(const umgap_taxonomy* tx) { return tx ? tx->len : 0; }
uint32_t umgap_taxonomy_reachable(const umgap_taxonomy* tx) { return tx ? tx->n_reach : 0; }

int umgap_taxonomy_snapping(const umgap_taxonomy* tx, int ranked_only, uint32_t* out) {
    if (!tx || !out) return fail(UMGAP_ERR_INVALID, "umgap_taxonomy_snapping: null argument");
    const std::vector<uint32_t>& s = tx->h_snap[ranked_only ? 1 : 0];
    memcpy(out, s.data(), s.size() * sizeof(uint32_t));
    return UMGAP_OK;
}

int umgap_taxonomy_preorder(const umgap_taxonomy* tx, uint32_t* pre, uint32_t* last, uint32_t* depth) {
    if (!tx) return fail(UMGAP_ERR_INVALID, "umgap_taxonomy_preorder: null argument");
    if (pre) memcpy(pre, tx->h_pre.data(), tx->len * sizeof(uint32_t));
    if (last) memcpy(last, tx->h_last.data(), tx->len * sizeof(uint32_t));
    if (depth) memcpy(depth, tx->h_depth.data(), tx->len * sizeof(uint32_t));
    return UMGAP_OK;
}

}  // extern "C"
