// aggregate.hip — per-read consensus taxon on the GPU (gfx950, wave64).
//
// Replaces the taxa2agg record loop (src/commands/taxa2agg.rs:161-180): agg::count / agg::filter
// (src/agg/mod.rs:27-44), the Aggregator implementations (tree::lca src/tree/lca.rs:32-40,
// tree::mix src/tree/mix.rs:43-64, rmq::lca src/rmq/lca.rs:60-90, rmq::rtl src/rmq/rtl.rs:39-57,
// rmq::mix src/rmq/mix.rs:57-93) and the snapping lookup (:178), over the preorder-interval
// taxonomy of taxonomy.hpp.  One wavefront owns one read at a time.
//
// Stage 1 (agg::count + agg::filter): the read's taxa sit one per lane (two registers cover 128
// taxa).  Unscored reads count through a 256-slot open-addressing hash table in LDS private to the
// wave: every lane inserts its taxon with ds_cmpst (the lane whose insert wins is the LEADER of
// that distinct taxon) and bumps the slot's counter with ds_add -- ~50 wave instructions per read
// whatever the number of distinct taxa.  Scored reads (-s) must add their scores in input order,
// so they use a ballot "peel" instead: v = taxon of the first unprocessed lane, ballot(t == v) =
// its copies, one iteration per DISTINCT taxon.
// Stage 2: leaders gather {pre, last} (the only taxonomy access that scales with the read).
// Stage 3 (LCA*, SURVEY.md §8a row 15, checked against the oracle's literal Tree::collapse):
//   M = surviving taxon with the largest preorder index (always a leaf of the induced tree);
//   h' = smallest-preorder surviving taxon that is NOT an ancestor-or-self of M (last[h] < pre[M]);
//   LCA* = M if there is no h', else LCA(h', M) = first node on M's parent chain with pre <= pre[h'].
#include "agg_common.hpp"
#include "bound.hpp"
#include "host_copy.hpp"

using namespace umgap;

namespace umgap {
int launch_general(const TaxView& V, int method, int strategy, float factor, float lower_bound, const uint32_t* d_taxa,
                   const float* d_scores, const uint64_t* d_offsets, uint32_t stride, size_t n_reads, uint32_t* d_out_taxon,
                   uint32_t* d_out_status, uint32_t* d_defer, uint32_t* d_big, hipStream_t stream, uint32_t* bound_codes = nullptr,
                   const uint32_t* d_id_of_pre = nullptr);
int launch_strided_group(const TaxView& V, int method, int strategy, float factor, float lower_bound, const uint32_t* d_taxa, uint32_t stride,
                         size_t n_reads, uint32_t* d_out_taxon, uint32_t* d_out_status, uint32_t* d_list, uint32_t* d_count, hipStream_t stream);
int launch_csr_lca_group(const TaxView& V, int method, float lower_bound, const uint32_t* d_taxa, const uint64_t* d_offsets, size_t n_reads, uint32_t* d_out_taxon,
                         uint32_t* d_out_status, uint32_t* d_list, uint32_t* d_count, hipStream_t stream, bool bound_hits);
void launch_deferred_codes_to_ids(const uint32_t* d_list, const uint32_t* d_count, const uint64_t* d_offsets, uint32_t* codes, const uint32_t* d_id_of_pre, hipStream_t stream);
}

namespace {
using namespace umgap::agg;

// ---------------------------------------------------------------------------------------------
// Stage 1 for reads of at most 128 taxa: t0/t1 = the lane's taxa (0 = none).  On return keep0 /
// keep1 say whether the lane is the leader of a surviving distinct taxon, cnt0/cnt1 its sum.
// Scores are summed in input order exactly like agg::count (src/agg/mod.rs:32-34).
// ---------------------------------------------------------------------------------------------
template <bool SCORED>
__device__ __forceinline__ void peel(uint32_t t0, uint32_t t1, float s0, float s1, float lower_bound, uint32_t lane,
                                     bool& keep0, bool& keep1, float& cnt0, float& cnt1) {
    uint64_t rem0 = __ballot(t0 != 0), rem1 = __ballot(t1 != 0);
    keep0 = keep1 = false;
    cnt0 = cnt1 = 0.0f;
    while (rem0 | rem1) {
        const bool in0 = rem0 != 0;
        const uint32_t src = in0 ? (uint32_t)__builtin_ctzll(rem0) : (uint32_t)__builtin_ctzll(rem1);
        const uint32_t v = in0 ? __builtin_amdgcn_readlane(t0, src) : __builtin_amdgcn_readlane(t1, src);
        const uint64_t m0 = __ballot(t0 == v), m1 = __ballot(t1 == v);
        float sum;
        if (SCORED) {
            sum = 0.0f;
            for (uint64_t b = m0; b; b &= b - 1) sum = __fadd_rn(sum, __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(uint32_t, s0), (uint32_t)__builtin_ctzll(b))));
            for (uint64_t b = m1; b; b &= b - 1) sum = __fadd_rn(sum, __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(uint32_t, s1), (uint32_t)__builtin_ctzll(b))));
        } else {
            sum = (float)(__builtin_popcountll(m0) + __builtin_popcountll(m1));
        }
        if (sum >= lower_bound) {  // agg::filter keeps freq >= lower_bound
            if (in0) { if (lane == src) { keep0 = true; cnt0 = sum; } }
            else     { if (lane == src) { keep1 = true; cnt1 = sum; } }
        }
        rem0 &= ~m0;
        rem1 &= ~m1;
    }
}

// ---------------------------------------------------------------------------------------------
// Generic slow path for reads longer than 128 taxa: O(n^2) count with scalar loads; every lane
// classifies the taxa at positions lane, lane+64, ...; only the FIRST occurrence of a taxon leads.
// ---------------------------------------------------------------------------------------------
template <bool SCORED>
__device__ __forceinline__ bool survives_generic(const uint32_t* __restrict__ rp, const float* __restrict__ sp, uint32_t n,
                                                 uint32_t j, uint32_t t, float lower_bound, float& sum) {
    if (t == 0) return false;
    sum = 0.0f;
    uint32_t c = 0;
    bool first = true;
    for (uint32_t i = 0; i < n; i++) {
        uint32_t ti = rp[i];
        if (ti == t) {
            if (i < j) first = false;
            if (SCORED) sum = __fadd_rn(sum, sp[i]); else c++;
        }
    }
    if (!SCORED) sum = (float)c;
    return first && sum >= lower_bound;
}

// LCA* over an arbitrary number of taxa; `for_each` visits every surviving leader hit.
// ---------------------------------------------------------------------------------------------
// LCA* kernel (both -m tree and -m rmq), one wave per read.
// ---------------------------------------------------------------------------------------------
// LISTED: only the reads on the list todo[0 .. *todo_count) -- the ones the group kernel of aggregate_group.hip left to this one
template <bool SCORED, bool LISTED = false>
__global__ __launch_bounds__(256) void k_agg_lca(TaxView V, int method, const uint32_t* __restrict__ taxa,
                                                 const float* __restrict__ scores, const uint64_t* __restrict__ offsets,
                                                 uint32_t stride, size_t n_reads, float lower_bound,
                                                 uint32_t* __restrict__ out_taxon, uint32_t* __restrict__ out_status,
                                                 const uint32_t* __restrict__ todo = nullptr, const uint32_t* __restrict__ todo_count = nullptr) {
    const uint32_t lane = threadIdx.x & 63;
    const size_t wave0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
    const bool tree_like = method != UMGAP_METHOD_RMQ;
    if (LISTED) n_reads = (size_t)*todo_count;
    __shared__ uint32_t lds_keys[4][256], lds_cnt[4][256];
    uint32_t* wkeys = lds_keys[threadIdx.x >> 6];
    uint32_t* wcnt = lds_cnt[threadIdx.x >> 6];
    if (!SCORED) {
#pragma unroll
        for (int i = 0; i < 4; i++) { wkeys[lane + 64 * i] = 0u; wcnt[lane + 64 * i] = 0u; }
    }
    for (size_t r_it = wave0; r_it < n_reads; r_it += nwaves) {
        const size_t r = LISTED ? (size_t)todo[r_it] : r_it;
        uint64_t off, end;
        if (offsets) { off = offsets[r]; end = offsets[r + 1]; } else { off = (uint64_t)r * stride; end = off + stride; }
        uint32_t off_lo = __builtin_amdgcn_readfirstlane((uint32_t)off), off_hi = __builtin_amdgcn_readfirstlane((uint32_t)(off >> 32));
        off = ((uint64_t)off_hi << 32) | off_lo;
        const uint32_t n = __builtin_amdgcn_readfirstlane((uint32_t)(end - off));
        const uint32_t* rp = taxa + off;
        const float* sp = SCORED ? scores + off : nullptr;

        uint64_t errkey = NOERR;
        Hit h0{NONE, NONE}, h1{NONE, NONE};
        uint32_t maxpre = 0, res_taxon, res_status;
        bool any = false;
        if (n <= 128) {
            // ---- fast path: whole read in two registers per lane ----
            uint32_t t0 = lane < n ? rp[lane] : 0u, t1 = 64 + lane < n ? rp[64 + lane] : 0u;
            float s0 = 0.0f, s1 = 0.0f;
            if (SCORED) { s0 = lane < n ? sp[lane] : 0.0f; s1 = 64 + lane < n ? sp[64 + lane] : 0.0f; }
            bool keep0, keep1;
            float c0, c1;
            if (SCORED) peel<SCORED>(t0, t1, s0, s1, lower_bound, lane, keep0, keep1, c0, c1);
            else lds_count(wkeys, wcnt, t0, t1, lower_bound, keep0, keep1, c0, c1);
            if (keep0) h0 = resolve(V, tree_like, t0, errkey);
            if (keep1) h1 = resolve(V, tree_like, t1, errkey);
            if (h0.pre != NONE) { any = true; maxpre = h0.pre; }
            if (h1.pre != NONE) { any = true; maxpre = max(maxpre, h1.pre); }
        } else {
            for (uint32_t c = 0; c * 64 < n; c++) {
                uint32_t j = c * 64 + lane;
                uint32_t t = j < n ? rp[j] : 0u;
                float sum;
                if (survives_generic<SCORED>(rp, sp, n, j, t, lower_bound, sum)) {
                    Hit h = resolve(V, tree_like, t, errkey);
                    if (h.pre != NONE) { any = true; maxpre = max(maxpre, h.pre); }
                }
            }
        }
        if (__any(errkey != NOERR)) {
            uint64_t ek = wave_min64(errkey);
            res_status = ((ek >> 32) & 1) ? UMGAP_READ_PANIC : UMGAP_READ_UNKNOWN_TAXON;
            res_taxon = (uint32_t)ek;
        } else if (!__any(any)) {
            res_status = UMGAP_READ_EMPTY;  // literal "1" (src/commands/taxa2agg.rs:174-175)
            res_taxon = 1;
        } else {
            const uint32_t M = wave_max(any ? maxpre : 0u);
            uint32_t hp = NONE;
            if (n <= 128) {
                if (h0.pre != NONE && h0.last < M) hp = h0.pre;
                if (h1.pre != NONE && h1.last < M) hp = min(hp, h1.pre);
            } else {
                for (uint32_t c = 0; c * 64 < n; c++) {  // long reads: classify again (rare)
                    uint32_t j = c * 64 + lane;
                    uint32_t t = j < n ? rp[j] : 0u;
                    float sum;
                    uint64_t dummy = NOERR;
                    if (survives_generic<SCORED>(rp, sp, n, j, t, lower_bound, sum)) {
                        Hit h = resolve(V, tree_like, t, dummy);
                        if (h.pre != NONE && h.last < M) hp = min(hp, h.pre);
                    }
                }
            }
            hp = wave_min(hp);
            if (hp != NONE) {
                // LCA(h', M) = parent of the shallowest node in (pre[h'], pre[M]]; its snap is precomputed
                uint32_t c = wave_argmin_depth(V, hp + 1, M, lane);
                res_taxon = V.psnap_of_pre[c];
            } else {
                res_taxon = V.snap_of_pre[M];  // snapping[aggregate] (src/commands/taxa2agg.rs:178)
            }
            res_status = res_taxon == NONE ? UMGAP_READ_PANIC : UMGAP_READ_OK;
        }
        if (lane == 0) {
            out_taxon[r] = res_taxon;
            out_status[r] = res_status;
        }
    }
}

// interval codes (bound.hpp) -> taxon ids, in place; 0 = no hit stays
__global__ __launch_bounds__(256) void k_all_codes_to_ids(uint32_t* __restrict__ hits, uint64_t n, const uint32_t* __restrict__ id_of_pre) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t step = (uint64_t)gridDim.x * blockDim.x;
    for (; i < n; i += step) {
        const uint32_t v = hits[i];
        if (v) hits[i] = id_of_pre[umgap::bound::pre_of(v)];
    }
}

inline int agg_grid(size_t n_reads) {
    size_t g = (n_reads + 3) / 4;
    if (g < 1) g = 1;
    if (g > 256 * 8) g = 256 * 8;
    return (int)g;
}

// bound_codes (umgap::aggregate_csr_bound_dev): the hits are interval codes of a table bound to tx, writable.  n_hits (0 = not known): how
// many hits the CSR reads hold together
int launch(umgap_taxonomy* tx, int method, int strategy, float factor, float lower_bound, int ranked_only,
           const uint32_t* d_taxa, const float* d_scores, const uint64_t* d_offsets, uint32_t stride, size_t n_reads,
           uint32_t* d_out_taxon, uint32_t* d_out_status, hipStream_t stream, uint32_t* bound_codes = nullptr, uint64_t n_hits = 0) {
    if (method != UMGAP_METHOD_TREE && method != UMGAP_METHOD_RMQ) return fail(UMGAP_ERR_INVALID, "unknown method");
    if (strategy < 0 || strategy > 2) return fail(UMGAP_ERR_INVALID, "unknown strategy");
    if (method == UMGAP_METHOD_TREE && strategy == UMGAP_STRATEGY_MRTL)
        return fail(UMGAP_ERR_INVALID, "Invalid invocation: Tree and MaximumRootToLeafPath cannot be combined");  // src/commands/taxa2agg.rs:134-138
    if (!n_reads) return UMGAP_OK;
    TaxView V = view_of(tx, ranked_only);
    dim3 grid(agg_grid(n_reads)), block(256);
    // (LCA*: the one-read-per-wave kernel below already spends a single round trip per stage; the group form of
    //  aggregate_group.hip measured 5 % slower on 16-hit reads, so it stays with the strategies that gain 2-3x from it)
    // LCA* over CSR reads that are SHORT on average (the caller knows their hits: at most eight a read -- a tryptic pipeline's): the
    // group kernel of aggregate_group.hip takes 32-64 reads per set of taxonomy round trips instead of one (10 M four-hit reads: 3.5 ms
    // one read per wave); reads of more than 128 hits are left on a list for the per-read kernel behind it.
    bool lca_group_path = strategy == UMGAP_STRATEGY_LCA_STAR && !d_scores && d_offsets && n_hits && n_hits <= 8 * (uint64_t)n_reads && n_reads < 0xFFFFFFFFull;
    if (const char* e = getenv("UMGAP_AGG_LCA_GROUP")) lca_group_path = strategy == UMGAP_STRATEGY_LCA_STAR && !d_scores && d_offsets && n_reads < 0xFFFFFFFFull && atoi(e) != 0;   // (tests, A/B)
    if (lca_group_path) {
        uint32_t* d_defer = defer_list_for(tx, stream, (n_reads + 1) * sizeof(uint32_t));
        if (!d_defer) return UMGAP_ERR_NOMEM;
        UMGAP_HIP(hipMemsetAsync(d_defer, 0, sizeof(uint32_t), stream));
        int rc = launch_csr_lca_group(V, method, lower_bound, d_taxa, d_offsets, n_reads, d_out_taxon, d_out_status, d_defer + 1, d_defer, stream, bound_codes != nullptr);
        if (rc) return rc;
        if (bound_codes) launch_deferred_codes_to_ids(d_defer + 1, d_defer, d_offsets, bound_codes, (const uint32_t*)tx->d_id_of_pre, stream);
        hipLaunchKernelGGL((k_agg_lca<false, true>), dim3(256), block, 0, stream, V, method, d_taxa, d_scores, d_offsets, stride, n_reads, lower_bound, d_out_taxon, d_out_status,
                           (const uint32_t*)(d_defer + 1), (const uint32_t*)d_defer);
    } else if (strategy == UMGAP_STRATEGY_LCA_STAR) {
        // (the per-read LCA* kernel knows taxon ids only: bound hits are turned back first)
        if (bound_codes && n_hits) hipLaunchKernelGGL(k_all_codes_to_ids, dim3((unsigned)std::min<uint64_t>((n_hits + 255) / 256, 256 * 8)), dim3(256), 0, stream, bound_codes, n_hits, (const uint32_t*)tx->d_id_of_pre);
        if (d_scores) hipLaunchKernelGGL(k_agg_lca<true>, grid, block, 0, stream, V, method, d_taxa, d_scores, d_offsets, stride, n_reads, lower_bound, d_out_taxon, d_out_status);
        else hipLaunchKernelGGL(k_agg_lca<false>, grid, block, 0, stream, V, method, d_taxa, d_scores, d_offsets, stride, n_reads, lower_bound, d_out_taxon, d_out_status);
    } else {
        uint32_t* d_defer = defer_list_for(tx, stream, 2 * (n_reads + 1) * sizeof(uint32_t));
        uint32_t* d_big = big_scratch_for(tx);
        if (!d_defer || !d_big) return UMGAP_ERR_NOMEM;
        return launch_general(V, method, strategy, factor, lower_bound, d_taxa, d_scores, d_offsets, stride, n_reads, d_out_taxon,
                              d_out_status, d_defer, d_big, stream, bound_codes, tx->d_id_of_pre);
    }
    UMGAP_HIP(hipGetLastError());
    return UMGAP_OK;
}

}  // extern "C"

namespace umgap {
// umgap_aggregate_dev for unscored CSR reads whose n_hits hits are the interval codes of a table BOUND to tx (tryptic.hip); the codes may
// be rewritten (to taxon ids, where a pass needs those)
int aggregate_csr_bound_dev(umgap_taxonomy* tx, int method, int strategy, float factor, float lower_bound, int ranked_only, uint32_t* d_codes,
                            uint64_t n_hits, const uint64_t* d_read_offsets, size_t n_reads, uint32_t* d_out_taxon, uint32_t* d_out_status, hipStream_t stream) {
    return launch(tx, method, strategy, factor, lower_bound, ranked_only, d_codes, nullptr, d_read_offsets, 0, n_reads, d_out_taxon, d_out_status, stream, d_codes, n_hits);
}
// umgap_aggregate_dev for unscored CSR reads by a caller that knows how many hits they hold together (tryptic.hip)
int aggregate_csr_counted_dev(umgap_taxonomy* tx, int method, int strategy, float factor, float lower_bound, int ranked_only, const uint32_t* d_taxa,
                              uint64_t n_hits, const uint64_t* d_read_offsets, size_t n_reads, uint32_t* d_out_taxon, uint32_t* d_out_status, hipStream_t stream) {
    return launch(tx, method, strategy, factor, lower_bound, ranked_only, d_taxa, nullptr, d_read_offsets, 0, n_reads, d_out_taxon, d_out_status, stream, nullptr, n_hits);
}
}  // namespace umgap

extern "C" {

int umgap_aggregate_dev(umgap_taxonomy* tx, int method, int strategy, float factor, float lower_bound, int ranked_only,
                        const uint32_t* d_taxa, const float* d_scores, const uint64_t* d_read_offsets, size_t n_reads,
                        uint32_t* d_out_taxon, uint32_t* d_out_status, void* stream) {
    if (!tx || (n_reads && (!d_read_offsets || !d_out_taxon || !d_out_status))) return fail(UMGAP_ERR_INVALID, "umgap_aggregate_dev: null argument");
    int rc = use_device(tx->device);
    if (rc) return rc;
    return launch(tx, method, strategy, factor, lower_bound, ranked_only, d_taxa, d_scores, d_read_offsets, 0, n_reads, d_out_taxon, d_out_status, (hipStream_t)stream);
}

int umgap_aggregate_strided_dev(umgap_taxonomy* tx, int method, int strategy, float factor, float lower_bound, int ranked_only,
                                const uint32_t* d_taxa, uint32_t stride, size_t n_reads, uint32_t* d_out_taxon,
                                uint32_t* d_out_status, void* stream) {
    if (!tx || (n_reads && (!d_taxa || !d_out_taxon || !d_out_status))) return fail(UMGAP_ERR_INVALID, "umgap_aggregate_strided_dev: null argument");
    int rc = use_device(tx->device);
    if (rc) return rc;
    return launch(tx, method, strategy, factor, lower_bound, ranked_only, d_taxa, nullptr, nullptr, stride, n_reads, d_out_taxon, d_out_status, (hipStream_t)stream);
}

}  // extern "C"
namespace {
__global__ __launch_bounds__(256) void k_rebase_read_offsets(uint64_t* __restrict__ off, size_t n, uint64_t base) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += step) off[i] -= base;
}
}  // namespace
extern "C" {

// Host arrays in, host arrays out, the link included: chunks of whole reads (at most 8 M taxa / 1 M reads) through three staging
// slots, an upload lane and a compute lane tied by events per slot (records.hip tells why the lanes are created together); the
// caller's arrays are copied from / to directly (pin them with umgap_host_alloc for the full rate of the link).
int umgap_aggregate(umgap_taxonomy* tx, int method, int strategy, float factor, float lower_bound, int ranked_only,
                    const uint32_t* taxa, const float* scores, const uint64_t* read_offsets, size_t n_reads,
                    uint32_t* out_taxon, uint32_t* out_status) {
    if (!tx || (n_reads && (!read_offsets || !out_taxon || !out_status))) return fail(UMGAP_ERR_INVALID, "umgap_aggregate: null argument");
    int rc = use_device(tx->device);
    if (rc) return rc;
    if (!n_reads) return UMGAP_OK;
    const uint64_t base = read_offsets[0], total = read_offsets[n_reads] - base;
    if (total && !taxa) return fail(UMGAP_ERR_INVALID, "umgap_aggregate: null taxa");
    constexpr int NS = 3;
    constexpr uint64_t CAPT = 8u << 20;     // taxa per chunk (a single read may exceed it: it is a chunk of its own)
    constexpr size_t CAPR = 1u << 20;       // reads per chunk
    // chunk plan: whole reads
    std::vector<size_t> cuts{0};
    uint64_t max_t = 0;
    size_t max_r = 0;
    while (cuts.back() < n_reads) {
        const size_t r0 = cuts.back();
        size_t lo = r0 + 1, hi = std::min(n_reads, r0 + CAPR);
        const uint64_t t0 = read_offsets[r0];
        while (lo < hi) {
            const size_t mid = lo + (hi - lo + 1) / 2;
            if (read_offsets[mid] >= t0 && read_offsets[mid] - t0 <= CAPT) lo = mid; else hi = mid - 1;
        }
        if (read_offsets[lo] < t0) return fail(UMGAP_ERR_INVALID, "umgap_aggregate: read_offsets must not decrease");
        max_t = std::max(max_t, read_offsets[lo] - t0);
        max_r = std::max(max_r, lo - r0);
        cuts.push_back(lo);
    }
    // device staging per slot: taxa | scores | offsets | out_taxon | out_status
    const size_t b_taxa = (max_t * 4 + 255) & ~(size_t)255, b_sc = scores ? b_taxa : 0, b_off = ((max_r + 1) * 8 + 255) & ~(size_t)255, b_out = (max_r * 4 + 255) & ~(size_t)255;
    const size_t per_slot = b_taxa + b_sc + b_off + 2 * b_out, need = NS * per_slot;
    if (need > tx->d_stage_bytes) {
        if (tx->d_stage) (void)hipFree(tx->d_stage);
        tx->d_stage = nullptr;
        tx->d_stage_bytes = 0;
        hipError_t e = hipMalloc(&tx->d_stage, need);
        if (e != hipSuccess) return fail(UMGAP_ERR_NOMEM, std::string("hipMalloc(aggregate staging): ") + hipGetErrorString(e));
        tx->d_stage_bytes = need;
    }
    if (int lanes = ensure_lanes(tx)) return lanes;
    const hipStream_t up = tx->stream2, run = tx->stream3;
    // pageable arrays travel through pinned staging with a threaded memcpy, both ways (records.hip tells why)
    auto is_pinned = [](const void* p0, const void* p1) {
        hipPointerAttribute_t a{};
        for (const void* p : {p0, p1}) {
            if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
            if (a.type != hipMemoryTypeHost) return false;
        }
        return true;
    };
    const bool in_pinned = (!total || (is_pinned(taxa + base, taxa + base + total - 1) && (!scores || is_pinned(scores + base, scores + base + total - 1)))) &&
                           is_pinned(read_offsets, read_offsets + n_reads);
    const bool out_pinned = is_pinned(out_taxon, out_taxon + n_reads - 1) && is_pinned(out_status, out_status + n_reads - 1);
    const size_t h_per_slot = (in_pinned ? 0 : b_taxa + b_sc + b_off) + (out_pinned ? 0 : 2 * b_out);
    if (NS * h_per_slot > tx->h_stage_bytes) {
        if (tx->h_stage) (void)hipHostFree(tx->h_stage);
        tx->h_stage = nullptr;
        tx->h_stage_bytes = 0;
        if (hipHostMalloc(&tx->h_stage, NS * h_per_slot, hipHostMallocDefault) != hipSuccess) return fail(UMGAP_ERR_NOMEM, "umgap_aggregate: hipHostMalloc of the staging failed");
        tx->h_stage_bytes = NS * h_per_slot;
    }
    struct HostSlot { uint32_t* taxa = nullptr; float* sc = nullptr; uint64_t* off = nullptr; uint32_t *ot = nullptr, *os = nullptr; } H[NS];
    for (int k = 0; k < NS; k++) {
        char* p = (char*)tx->h_stage + (size_t)k * h_per_slot;
        if (!in_pinned) { H[k].taxa = (uint32_t*)p; p += b_taxa; if (scores) { H[k].sc = (float*)p; p += b_sc; } H[k].off = (uint64_t*)p; p += b_off; }
        if (!out_pinned) { H[k].ot = (uint32_t*)p; p += b_out; H[k].os = (uint32_t*)p; }
    }
    size_t out_at[NS] = {}, out_n[NS] = {};
    auto drain_out = [&](int k) {
        if (!out_n[k]) return;
        memcpy(out_taxon + out_at[k], H[k].ot, out_n[k] * 4);
        memcpy(out_status + out_at[k], H[k].os, out_n[k] * 4);
        out_n[k] = 0;
    };
    struct Events {
        hipEvent_t uploaded[NS] = {}, done[NS] = {};
        ~Events() { for (hipEvent_t x : uploaded) if (x) (void)hipEventDestroy(x); for (hipEvent_t x : done) if (x) (void)hipEventDestroy(x); }
    } ev;
    for (int i = 0; i < NS; i++) {
        UMGAP_HIP(hipEventCreateWithFlags(&ev.uploaded[i], hipEventDisableTiming));
        UMGAP_HIP(hipEventCreateWithFlags(&ev.done[i], hipEventDisableTiming));
    }
    for (size_t c = 0; c + 1 < cuts.size() && !rc; c++) {
        const int slot = (int)(c % NS);
        const size_t r0 = cuts[c], nr = cuts[c + 1] - r0;
        const uint64_t t0 = read_offsets[r0], nt = read_offsets[cuts[c + 1]] - t0;
        char* p = (char*)tx->d_stage + (size_t)slot * per_slot;
        uint32_t* d_taxa = (uint32_t*)p; p += b_taxa;
        float* d_sc = scores ? (float*)p : nullptr; p += b_sc;
        uint64_t* d_off = (uint64_t*)p; p += b_off;
        uint32_t* d_ot = (uint32_t*)p; p += b_out;
        uint32_t* d_os = (uint32_t*)p;
        if (c >= (size_t)NS) {                 // the slot's previous tenant is through (its results have left)
            UMGAP_HIP_BREAK(hipEventSynchronize(ev.done[slot]));
            drain_out(slot);
        }
        const uint32_t* src_t = taxa + t0;
        const float* src_s = scores ? scores + t0 : nullptr;
        const uint64_t* src_o = read_offsets + r0;
        if (!in_pinned) {
            if (nt) par_memcpy(H[slot].taxa, src_t, nt * 4);
            if (scores && nt) par_memcpy(H[slot].sc, src_s, nt * 4);
            par_memcpy(H[slot].off, src_o, (nr + 1) * 8);
            src_t = H[slot].taxa; src_s = H[slot].sc; src_o = H[slot].off;
        }
        if (nt) UMGAP_HIP_BREAK(hipMemcpyAsync(d_taxa, src_t, nt * 4, hipMemcpyHostToDevice, up));
        if (scores && nt) UMGAP_HIP_BREAK(hipMemcpyAsync(d_sc, src_s, nt * 4, hipMemcpyHostToDevice, up));
        UMGAP_HIP_BREAK(hipMemcpyAsync(d_off, src_o, (nr + 1) * 8, hipMemcpyHostToDevice, up));
        UMGAP_HIP_BREAK(hipEventRecord(ev.uploaded[slot], up));
        UMGAP_HIP_BREAK(hipStreamWaitEvent(run, ev.uploaded[slot], 0));
        hipLaunchKernelGGL(k_rebase_read_offsets, dim3((unsigned)std::min<size_t>((nr + 256) / 256, 1024)), dim3(256), 0, run, d_off, nr + 1, t0);
        rc = launch(tx, method, strategy, factor, lower_bound, ranked_only, d_taxa, d_sc, d_off, 0, nr, d_ot, d_os, run, nullptr, nt);
        if (rc) break;
        UMGAP_HIP_BREAK(hipMemcpyAsync(out_pinned ? out_taxon + r0 : H[slot].ot, d_ot, nr * 4, hipMemcpyDeviceToHost, run));
        UMGAP_HIP_BREAK(hipMemcpyAsync(out_pinned ? out_status + r0 : H[slot].os, d_os, nr * 4, hipMemcpyDeviceToHost, run));
        if (!out_pinned) { out_at[slot] = r0; out_n[slot] = nr; }
        UMGAP_HIP_BREAK(hipEventRecord(ev.done[slot], run));
    }
    const hipError_t e0 = hipStreamSynchronize(up), e1 = hipStreamSynchronize(run);
    if (!rc && e0 == hipSuccess && e1 == hipSuccess) for (int k = 0; k < NS; k++) drain_out(k);
    if (rc) return rc;
    if (e0 != hipSuccess || e1 != hipSuccess) return fail(UMGAP_ERR_DEVICE, std::string("umgap_aggregate: ") + hipGetErrorString(e0 != hipSuccess ? e0 : e1));
    return UMGAP_OK;
}

}  // extern "C"
