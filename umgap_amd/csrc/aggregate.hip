// aggregate.hip — per-read consensus taxon on the GPU (gfx950, wave64).
//
// Replaces the taxa2agg record loop (src/commands/taxa2agg.rs:161-180): agg::count / agg::filter
// (src/agg/mod.rs:27-44), the Aggregator implementations (tree::lca src/tree/lca.rs:32-40,
// tree::mix src/tree/mix.rs:43-64, rmq::lca src/rmq/lca.rs:60-90, rmq::rtl src/rmq/rtl.rs:39-57,
// rmq::mix src/rmq/mix.rs:57-93) and the snapping lookup (:178), over the preorder-interval
// taxonomy of taxonomy.hpp.  One wavefront owns one read at a time.
//
// LCA* closed form (checked against the oracle's literal Tree::collapse):
//   D = distinct taxa of the read that survive the filters;  M = the one with the largest preorder
//   index (always a leaf of the induced tree);  h' = the smallest-preorder member of D that is NOT
//   an ancestor-or-self of M (last[h] < pre[M]).  LCA* = M if h' does not exist, else LCA(h', M)
//   = the first node on M's parent chain whose preorder index is <= pre[h'].
#include "taxonomy.hpp"

using namespace umgap;

namespace {
constexpr uint32_t NONE = UMGAP_NONE;

__device__ __forceinline__ uint32_t wave_max(uint32_t v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { uint32_t w = (uint32_t)__shfl_xor((int)v, o, 64); v = w > v ? w : v; }
    return v;
}
__device__ __forceinline__ uint32_t wave_min(uint32_t v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { uint32_t w = (uint32_t)__shfl_xor((int)v, o, 64); v = w < v ? w : v; }
    return v;
}
__device__ __forceinline__ uint64_t wave_min64(uint64_t v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, o, 64);
        uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), o, 64);
        uint64_t w = ((uint64_t)hi << 32) | lo;
        v = w < v ? w : v;
    }
    return v;
}

// Sum of the scores of taxon `t` over the read, accumulated in input order exactly like
// agg::count (src/agg/mod.rs:32-34: `*entry += count`, starting from 0.0).  Unscored input
// counts 1.0 per occurrence (src/commands/taxa2agg.rs:150-152); integer counts are exact in f32.
// `rp`/`sp` and `n` are wave-uniform, so the loads are scalar (s_load) and shared by all lanes.
template <bool SCORED>
__device__ __forceinline__ float read_sum(const uint32_t* __restrict__ rp, const float* __restrict__ sp, uint32_t n,
                                          uint32_t t) {
    if (SCORED) {
        float acc = 0.0f;
        for (uint32_t j = 0; j < n; j++) {
            uint32_t tj = rp[j];
            float sj = sp[j];
            if (tj == t) acc = __fadd_rn(acc, sj);
        }
        return acc;
    } else {
        uint32_t c = 0;
        uint32_t j = 0;
        for (; j + 4 <= n; j += 4) {
            uint32_t a = rp[j], b = rp[j + 1], cc = rp[j + 2], d = rp[j + 3];
            c += (a == t) + (b == t) + (cc == t) + (d == t);
        }
        for (; j < n; j++) c += (rp[j] == t);
        return (float)c;
    }
}

struct Hit {
    uint32_t pre, last;  // pre == NONE: this lane holds no surviving hit
};

// Classify one candidate taxon of the read: survives the filters?  known?  Returns the hit and
// folds failures into `errkey` (smaller = reported first: BFS level, then Unknown before panic,
// then payload — the order tree::Tree::new would meet them, src/tree/mod.rs:35-46).
template <bool SCORED>
__device__ __forceinline__ Hit classify(const TaxView& V, int method, const uint32_t* rp, const float* sp, uint32_t n,
                                        uint32_t t, float lower_bound, bool counting, uint64_t& errkey) {
    Hit h{NONE, NONE};
    if (t == 0) return h;  // src/commands/taxa2agg.rs:169 drops taxon 0
    if (counting) {
        float s = read_sum<SCORED>(rp, sp, n, t);
        if (!(s >= lower_bound)) return h;  // agg::filter keeps freq >= lower_bound
    }
    if (t >= V.len) {
        // tree/rtl: parents[id] out of bounds => panic; rmq::lca/mix: HashMap miss => UnknownTaxon(id)
        errkey = min(errkey, method == UMGAP_METHOD_RMQ ? (uint64_t)t : ((1ull << 32) | t));
        return h;
    }
    uint2 pl = V.by_id[t];
    if (pl.x == NONE) {
        if (method == UMGAP_METHOD_RMQ) {
            errkey = min(errkey, (uint64_t)t);  // first_occurence(t) fails (src/rmq/lca.rs:49-54)
        } else {
            uint64_t k = ((uint64_t)V.err_dist[t] << 33) | ((uint64_t)(V.err_kind[t] == 2) << 32) | V.err_payload[t];
            errkey = min(errkey, k);
        }
        return h;
    }
    h.pre = pl.x;
    h.last = pl.y;
    return h;
}

// ---------------------------------------------------------------------------------------------
// LCA* (both -m tree and -m rmq), one wave per read.
// ---------------------------------------------------------------------------------------------
template <bool SCORED>
__global__ __launch_bounds__(256) void k_agg_lca(TaxView V, int method, const uint32_t* __restrict__ taxa,
                                                 const float* __restrict__ scores, const uint64_t* __restrict__ offsets,
                                                 uint32_t stride, size_t n_reads, float lower_bound,
                                                 uint32_t* __restrict__ out_taxon, uint32_t* __restrict__ out_status) {
    const uint32_t lane = threadIdx.x & 63;
    const size_t wave0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
    // unscored input with lower_bound <= 1: every non-zero taxon has count >= 1 >= bound
    const bool counting = SCORED || !(lower_bound <= 1.0f);
    for (size_t r = wave0; r < n_reads; r += nwaves) {
        uint64_t off, end;
        if (offsets) { off = offsets[r]; end = offsets[r + 1]; } else { off = (uint64_t)r * stride; end = off + stride; }
        // make the read's base and length scalar so that read_sum's loads are s_loads
        uint32_t off_lo = __builtin_amdgcn_readfirstlane((uint32_t)off), off_hi = __builtin_amdgcn_readfirstlane((uint32_t)(off >> 32));
        off = ((uint64_t)off_hi << 32) | off_lo;
        const uint32_t n = __builtin_amdgcn_readfirstlane((uint32_t)(end - off));
        const uint32_t* rp = taxa + off;
        const float* sp = SCORED ? scores + off : nullptr;

        uint64_t errkey = ~0ull;
        Hit h0{NONE, NONE}, h1{NONE, NONE};
        uint32_t maxpre = 0;
        bool any = false;
        // phase 1: classify every taxon, remember the first two chunks in registers
        for (uint32_t c = 0; c * 64 < n; c++) {
            uint32_t j = c * 64 + lane;
            uint32_t t = j < n ? rp[j] : 0u;
            Hit h = classify<SCORED>(V, method, rp, sp, n, t, lower_bound, counting, errkey);
            if (c == 0) h0 = h;
            if (c == 1) h1 = h;
            if (h.pre != NONE) { any = true; maxpre = max(maxpre, h.pre); }
        }
        uint32_t res_taxon, res_status;
        uint64_t ek = wave_min64(errkey);
        if (ek != ~0ull) {
            bool panic = (ek >> 32) & 1;
            res_status = panic ? UMGAP_READ_PANIC : UMGAP_READ_UNKNOWN_TAXON;
            res_taxon = (uint32_t)ek;
        } else if (!__any(any)) {
            res_status = UMGAP_READ_EMPTY;  // literal "1" (src/commands/taxa2agg.rs:174-175)
            res_taxon = 1;
        } else {
            const uint32_t M = wave_max(any ? maxpre : 0u);
            // phase 2: smallest-preorder surviving taxon that is not an ancestor-or-self of M
            uint32_t hp = NONE;
            if (h0.pre != NONE && h0.last < M) hp = min(hp, h0.pre);
            if (h1.pre != NONE && h1.last < M) hp = min(hp, h1.pre);
            for (uint32_t c = 2; c * 64 < n; c++) {  // long reads: classify again (rare)
                uint32_t j = c * 64 + lane;
                uint32_t t = j < n ? rp[j] : 0u;
                uint64_t dummy = ~0ull;
                Hit h = classify<SCORED>(V, method, rp, sp, n, t, lower_bound, counting, dummy);
                if (h.pre != NONE && h.last < M) hp = min(hp, h.pre);
            }
            hp = wave_min(hp);
            uint32_t x = M;
            if (hp != NONE) {
                // LCA(h', M): climb M's parent chain (uniform => scalar loads)
                while (x > hp) x = __builtin_amdgcn_readfirstlane(V.parent_pre[x]);
            }
            res_taxon = V.snap_of_pre[x];  // snapping[aggregate] (src/commands/taxa2agg.rs:178)
            res_status = res_taxon == NONE ? UMGAP_READ_PANIC : UMGAP_READ_OK;
        }
        if (lane == 0) {
            out_taxon[r] = res_taxon;
            out_status[r] = res_status;
        }
    }
}

inline int agg_grid(size_t n_reads) {
    size_t g = (n_reads + 3) / 4;
    if (g < 1) g = 1;
    if (g > 256 * 8) g = 256 * 8;
    return (int)g;
}

int launch(umgap_taxonomy* tx, int method, int strategy, float factor, float lower_bound, int ranked_only,
           const uint32_t* d_taxa, const float* d_scores, const uint64_t* d_offsets, uint32_t stride, size_t n_reads,
           uint32_t* d_out_taxon, uint32_t* d_out_status, hipStream_t stream) {
    if (method != UMGAP_METHOD_TREE && method != UMGAP_METHOD_RMQ) return fail(UMGAP_ERR_INVALID, "unknown method");
    if (strategy < 0 || strategy > 2) return fail(UMGAP_ERR_INVALID, "unknown strategy");
    if (method == UMGAP_METHOD_TREE && strategy == UMGAP_STRATEGY_MRTL)
        return fail(UMGAP_ERR_INVALID, "Invalid invocation: Tree and MaximumRootToLeafPath cannot be combined");  // src/commands/taxa2agg.rs:134-138
    if (!n_reads) return UMGAP_OK;
    TaxView V = view_of(tx, ranked_only);
    dim3 grid(agg_grid(n_reads)), block(256);
    if (strategy == UMGAP_STRATEGY_LCA_STAR) {
        if (d_scores) hipLaunchKernelGGL(k_agg_lca<true>, grid, block, 0, stream, V, method, d_taxa, d_scores, d_offsets, stride, n_reads, lower_bound, d_out_taxon, d_out_status);
        else hipLaunchKernelGGL(k_agg_lca<false>, grid, block, 0, stream, V, method, d_taxa, d_scores, d_offsets, stride, n_reads, lower_bound, d_out_taxon, d_out_status);
    } else {
        return fail(UMGAP_ERR_INVALID, "strategy not implemented on the device yet");
    }
    UMGAP_HIP(hipGetLastError());
    (void)factor;
    return UMGAP_OK;
}

}  // namespace

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

int umgap_aggregate(umgap_taxonomy* tx, int method, int strategy, float factor, float lower_bound, int ranked_only,
                    const uint32_t* taxa, const float* scores, const uint64_t* read_offsets, size_t n_reads,
                    uint32_t* out_taxon, uint32_t* out_status) {
    if (!tx || (n_reads && (!read_offsets || !out_taxon || !out_status))) return fail(UMGAP_ERR_INVALID, "umgap_aggregate: null argument");
    int rc = use_device(tx->device);
    if (rc) return rc;
    if (!n_reads) return UMGAP_OK;
    const uint64_t base = read_offsets[0], total = read_offsets[n_reads] - base;
    if (total && !taxa) return fail(UMGAP_ERR_INVALID, "umgap_aggregate: null taxa");
    // device staging: taxa | scores | offsets | out_taxon | out_status
    size_t b_taxa = (total * 4 + 15) & ~(size_t)15, b_sc = scores ? b_taxa : 0, b_off = ((n_reads + 1) * 8 + 15) & ~(size_t)15,
           b_out = (n_reads * 4 + 15) & ~(size_t)15;
    size_t need = b_taxa + b_sc + b_off + 2 * b_out;
    if (need > tx->d_stage_bytes) {
        if (tx->d_stage) (void)hipFree(tx->d_stage);
        tx->d_stage = nullptr;
        tx->d_stage_bytes = 0;
        hipError_t e = hipMalloc(&tx->d_stage, need);
        if (e != hipSuccess) return fail(UMGAP_ERR_NOMEM, std::string("hipMalloc(aggregate staging): ") + hipGetErrorString(e));
        tx->d_stage_bytes = need;
    }
    char* p = (char*)tx->d_stage;
    uint32_t* d_taxa = (uint32_t*)p; p += b_taxa;
    float* d_sc = scores ? (float*)p : nullptr; p += b_sc;
    uint64_t* d_off = (uint64_t*)p; p += b_off;
    uint32_t* d_ot = (uint32_t*)p; p += b_out;
    uint32_t* d_os = (uint32_t*)p;
    std::vector<uint64_t> rel(n_reads + 1);
    for (size_t i = 0; i <= n_reads; i++) rel[i] = read_offsets[i] - base;
    if (total) UMGAP_HIP(hipMemcpyAsync(d_taxa, taxa + base, total * 4, hipMemcpyHostToDevice, tx->stream));
    if (scores && total) UMGAP_HIP(hipMemcpyAsync(d_sc, scores + base, total * 4, hipMemcpyHostToDevice, tx->stream));
    UMGAP_HIP(hipMemcpyAsync(d_off, rel.data(), (n_reads + 1) * 8, hipMemcpyHostToDevice, tx->stream));
    rc = launch(tx, method, strategy, factor, lower_bound, ranked_only, d_taxa, d_sc, d_off, 0, n_reads, d_ot, d_os, tx->stream);
    if (rc) { (void)hipStreamSynchronize(tx->stream); return rc; }
    UMGAP_HIP(hipMemcpyAsync(out_taxon, d_ot, n_reads * 4, hipMemcpyDeviceToHost, tx->stream));
    UMGAP_HIP(hipMemcpyAsync(out_status, d_os, n_reads * 4, hipMemcpyDeviceToHost, tx->stream));
    UMGAP_HIP(hipStreamSynchronize(tx->stream));
    return UMGAP_OK;
}

}  // extern "C"
