// agg_general.hpp -- the per-read body of the general aggregation (MRTL, tree-hybrid, rmq-hybrid), shared by
// k_agg_general (aggregate_general.hip: hits in HBM) and k_classify9_general (classify.hip: hits in LDS).
// Algorithms and reference citations: aggregate_general.hip.
#pragma once
#include "agg_common.hpp"
#include "bound.hpp"

namespace umgap {
namespace agg {

__device__ __forceinline__ void wave_sync() {  // LDS hand-off between lanes of one wave
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// The working set of one read: in LDS, sized at compile time ...
template <int CAP, int HCAP>
struct Scratch {
    uint32_t hkeys[HCAP], hcnt[HCAP];
    uint32_t a_tax[CAP], a_slot[CAP];
    float a_cnt[CAP];
    uint32_t s_pre[CAP], s_last[CAP];
    float s_cnt[CAP];
    uint32_t b_pre[CAP], b_last[CAP];
    float b_cnt[CAP];
    uint32_t b_dep[CAP], b_adj[CAP], b_lca[CAP];
    static constexpr bool kLanePerElement = CAP <= 64;
    __device__ static constexpr uint32_t cap() { return CAP; }
    __device__ static constexpr uint32_t hcap() { return HCAP; }
    __device__ static void sync() { wave_sync(); }
};
// ... or in global memory, sized at run time (the pass for reads with more distinct taxa than any LDS scratch holds: the
// reference has no limit, src/tree/mix.rs:43-64, src/rmq/rtl.rs:39-57).  Lanes of one wave hand data to each other through
// it, so a hand-off is an agent-scope fence (the vector cache is not coherent with the stores that go to L2).
struct ScratchGlobal {
    uint32_t *hkeys, *hcnt, *a_tax, *a_slot;
    float* a_cnt;
    uint32_t *s_pre, *s_last;
    float* s_cnt;
    uint32_t *b_pre, *b_last;
    float* b_cnt;
    uint32_t *b_dep, *b_adj, *b_lca;
    uint32_t cap_, hcap_;
    static constexpr bool kLanePerElement = false;
    __device__ uint32_t cap() const { return cap_; }
    __device__ uint32_t hcap() const { return hcap_; }
    __device__ static void sync() {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
        __builtin_amdgcn_wave_barrier();
    }
    // words of global memory one read's scratch takes
    __host__ __device__ static size_t words(uint32_t cap, uint32_t hcap) { return 2 * (size_t)hcap + 12 * (size_t)cap; }
    __host__ __device__ void carve(uint32_t* base, uint32_t cap, uint32_t hcap) {
        cap_ = cap; hcap_ = hcap;
        hkeys = base; base += hcap; hcnt = base; base += hcap;
        a_tax = base; base += cap; a_slot = base; base += cap; a_cnt = (float*)base; base += cap;
        s_pre = base; base += cap; s_last = base; base += cap; s_cnt = (float*)base; base += cap;
        b_pre = base; base += cap; b_last = base; base += cap; b_cnt = (float*)base; base += cap;
        b_dep = base; base += cap; b_adj = base; base += cap; b_lca = base;
    }
};
__device__ __forceinline__ uint64_t lanemask_lt(uint32_t lane) { return (1ull << lane) - 1ull; }
__device__ __forceinline__ float wave_maxf(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { float w = __shfl_xor(v, o, 64); v = w > v ? w : v; }
    return v;
}

// Packed (depth << 32 | position) of a shallowest node with preorder index in [lo, hi], lo <= hi, for ONE lane: the two
// 8-blocks at the ends are read whole (two aligned 16-byte loads each) and scanned in registers, the blocks in between
// come from the 8-block sparse table; every load is independent of the others (one round trip).  The first version walked
// a binary-lifting table: seven dependent gathers per LCA, as many random requests per read as the probes themselves.
typedef uint32_t agg_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint64_t lane_argmin_depth(const TaxView& V, uint32_t lo, uint32_t hi) {
    // Callers pass preorder indexes of resolved taxa only.  Two instructions of insurance all the same: an index past the tables
    // (NONE from an unresolved taxon that slipped through) would be a GPU memory fault, which on this hardware can reset the
    // node for everyone -- clamped, it is a wrong LCA in a read that is in an error state anyway.
    hi = min(hi, V.n_blocks8 * 8u - 1u);
    lo = min(lo, hi);
    const uint32_t bl = lo >> 3, bh = hi >> 3;
    const agg_u32x4* d4 = reinterpret_cast<const agg_u32x4*>(V.depth);
    const agg_u32x4 a0 = d4[2 * (size_t)bl], a1 = d4[2 * (size_t)bl + 1];
    agg_u32x4 b0 = a0, b1 = a1;
    if (bh != bl) { b0 = d4[2 * (size_t)bh]; b1 = d4[2 * (size_t)bh + 1]; }
    uint64_t e0 = ~0ull, e1 = ~0ull;
    if (bh > bl + 1) {
        const uint32_t len = bh - bl - 1, k = 31u - (uint32_t)__builtin_clz(len);
        e0 = V.sparse8[(size_t)k * V.n_blocks8 + bl + 1];
        e1 = V.sparse8[(size_t)k * V.n_blocks8 + bh - (1u << k)];
    }
    uint64_t best = e0 < e1 ? e0 : e1;
    const uint32_t da[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w}, db[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
    for (uint32_t i = 0; i < 8; i++) {
        const uint32_t pa = bl * 8 + i, pb = bh * 8 + i;
        const uint64_t ca = ((uint64_t)da[i] << 32) | pa, cb = ((uint64_t)db[i] << 32) | pb;
        if (pa >= lo && pa <= hi && ca < best) best = ca;
        if (pb >= lo && pb <= hi && cb < best) best = cb;
    }
    return best;
}

// One read through count -> filter -> resolve -> sort -> strategy, by one wave with its scratch S (hash table clean on
// entry and on exit).  rp / sp may point to global memory or to LDS (the fused kernel).  Returns false when the read has
// more than CAP distinct taxa (nothing written; S clean).
// BOUND: the taxa are the interval codes of a table bound to the taxonomy (bound.hpp): {pre, last} come from the code
// itself when the subtree is smaller than 32 nodes (exact size), from one 4-byte gather otherwise; nothing can be unknown.
template <bool SCORED, class SCR, bool BOUND = false>
__device__ __forceinline__ bool general_read_t(const TaxView& V, SCR& S, int method, int strategy, float factor, float lower_bound,
                                               const uint32_t* rp, const float* sp, uint64_t n, uint32_t lane, uint32_t& res_taxon, uint32_t& res_status) {
    const uint32_t CAP = S.cap(), HCAP = S.hcap();
    // rmq::rtl walks `ancestors` like the tree aggregators (src/rmq/rtl.rs:42-49)
    const bool tree_like = method != UMGAP_METHOD_RMQ || strategy == UMGAP_STRATEGY_MRTL;
    res_taxon = 0;
    res_status = UMGAP_READ_OK;
    bool done = false;
    {
        // ---- 1. distinct taxa (a_tax) with their slots -----------------------------------
        uint32_t nd = 0;
        bool overflow = false;
        for (uint64_t base = 0; base < n && !overflow; base += 64) {
            uint64_t j = base + lane;
            uint32_t t = j < n ? rp[j] : 0u;
            bool lead = false, full = false;
            uint32_t slot = 0;
            if (t != 0) {  // src/commands/taxa2agg.rs:169 drops taxon 0
                slot = (t * 0x9E3779B1u) >> (32 - __builtin_ctz(HCAP));
                uint32_t probes = 0;
                for (;;) {
                    uint32_t old = atomicCAS(&S.hkeys[slot], 0u, t);
                    if (old == 0u) { lead = true; break; }
                    if (old == t) break;
                    slot = (slot + 1) & (HCAP - 1);
                    if (++probes == HCAP) { full = true; break; }
                }
                if (!SCORED && !full) atomicAdd(&S.hcnt[slot], 1u);
            }
            uint64_t lm = __ballot(lead);
            uint32_t idx = nd + (uint32_t)__builtin_popcountll(lm & lanemask_lt(lane));
            if (lead && idx < CAP) { S.a_tax[idx] = t; S.a_slot[idx] = slot; }
            nd += (uint32_t)__builtin_popcountll(lm);
            overflow = nd > CAP || __any(full);
        }
        SCR::sync();
        if (overflow) {
            // give the table back clean; the caller hands the read to a pass with a larger scratch
            for (uint32_t i = lane; i < HCAP; i += 64) { S.hkeys[i] = 0u; S.hcnt[i] = 0u; }
            SCR::sync();
            return false;
        }
        // multiplicities, then clean the slots this read used
        for (uint32_t e = lane; e < nd; e += 64) {
            uint32_t slot = S.a_slot[e];
            if (!SCORED) S.a_cnt[e] = (float)S.hcnt[slot];
            S.hkeys[slot] = 0u;
            S.hcnt[slot] = 0u;
        }
        if (SCORED) {
            // scores are added in input order (agg::count): every lane scans the read for its taxa
            for (uint32_t eb = 0; eb < nd; eb += 64) {
                uint32_t e = eb + lane;
                uint32_t mine = e < nd ? S.a_tax[e] : 0u;
                float acc = 0.0f;
                for (uint64_t j = 0; j < n; j++) {
                    uint32_t tj = rp[j];
                    float sj = sp[j];
                    if (tj == mine) acc = __fadd_rn(acc, sj);
                }
                if (e < nd) S.a_cnt[e] = acc;
            }
        }
        SCR::sync();

        // ---- 2. filter + resolve + compact the survivors ---------------------------------
        uint32_t D = 0;
        uint64_t errkey = NOERR;
        for (uint32_t eb = 0; eb < nd; eb += 64) {
            uint32_t e = eb + lane;
            bool ok = e < nd && S.a_cnt[e] >= lower_bound;  // agg::filter
            Hit h{NONE, NONE};
            if (ok) {
                if (BOUND) {
                    uint32_t lo, hi;
                    bound::size_bounds(S.a_tax[e], lo, hi);
                    h.pre = bound::pre_of(S.a_tax[e]);
                    h.last = lo == hi ? h.pre + lo : V.last_of_pre[h.pre];
                } else h = resolve(V, tree_like, S.a_tax[e], errkey);
            }
            bool surv = ok && h.pre != NONE;
            uint64_t m = __ballot(surv);
            uint32_t idx = D + (uint32_t)__builtin_popcountll(m & lanemask_lt(lane));
            if (surv) { S.s_pre[idx] = h.pre; S.s_last[idx] = h.last; S.s_cnt[idx] = S.a_cnt[e]; }
            D += (uint32_t)__builtin_popcountll(m);
        }
        SCR::sync();
        if (__any(errkey != NOERR)) {
            uint64_t ek = wave_min64(errkey);
            res_status = ((ek >> 32) & 1) ? UMGAP_READ_PANIC : UMGAP_READ_UNKNOWN_TAXON;
            res_taxon = (uint32_t)ek;
            done = true;
        } else if (D == 0) {
            res_status = UMGAP_READ_EMPTY;  // literal "1" (src/commands/taxa2agg.rs:174-175)
            res_taxon = 1;
            done = true;
        }

        if (!done) {
            // ---- 3. sort by preorder index (distinct taxa => distinct keys): rank by counting ----
            for (uint32_t eb = 0; eb < D; eb += 64) {
                uint32_t e = eb + lane;
                uint32_t p = e < D ? S.s_pre[e] : 0u, rank = 0;
                for (uint32_t f = 0; f < D; f++) rank += S.s_pre[f] < p;
                if (e < D) { S.b_pre[rank] = p; S.b_last[rank] = S.s_last[e]; S.b_cnt[rank] = S.s_cnt[e]; }
            }
            SCR::sync();
            uint32_t ans = NONE;  // preorder index of the aggregate
            if (strategy == UMGAP_STRATEGY_MRTL) {
                // ---- MRTL ----
                float best = -__builtin_inff();
                uint32_t best_i = NONE;
                for (uint32_t ib = 0; ib < D; ib += 64) {
                    uint32_t i = ib + lane;
                    if (i < D) {
                        float score = S.b_cnt[i];
                        uint32_t pi = S.b_pre[i];
                        for (uint32_t j = i; j-- > 0;)
                            if (S.b_last[j] >= pi) score = __fadd_rn(score, S.b_cnt[j]);  // ancestor, nearest first
                        if (score > best) { best = score; best_i = i; }
                    }
                }
                float m = wave_maxf(best);
                uint32_t bi = wave_min(best == m ? best_i : NONE);
                ans = S.b_pre[bi];
            } else {
                // adjacent LCAs (both hybrids)
                for (uint32_t kb = 0; kb < D; kb += 64) {
                    uint32_t k = kb + lane;
                    uint32_t du = 0;
                    if (k < D) { du = V.depth[S.b_pre[k]]; S.b_dep[k] = du; }
                    if (k + 1 < D) {
                        const uint32_t u = S.b_pre[k], v = S.b_pre[k + 1];
                        if (S.b_last[k] >= v) { S.b_lca[k] = u; S.b_adj[k] = du; }      // u is an ancestor of v
                        else {   // LCA = parent of a shallowest node in (pre[u], pre[v]]; its depth is that node's minus one
                            const uint64_t m = lane_argmin_depth(V, u + 1, v);
                            S.b_lca[k] = V.parent_pre[(uint32_t)m];
                            S.b_adj[k] = (uint32_t)(m >> 32) - 1u;
                        }
                    }
                }
                SCR::sync();
                if (method == UMGAP_METHOD_TREE && SCR::kLanePerElement && !SCORED) {
                    // ---- tree::mix, one lane per list element: every level of the greedy descent is three wave
                    // reductions instead of two serial scans of the range.  Unscored counts are small integers, so
                    // the children's values (Tree::aggregate) are exact differences of one prefix sum.
                    const uint32_t k = lane;
                    float pfx = k < D ? S.b_cnt[k] : 0.0f;
#pragma unroll
                    for (int o = 1; o < 64; o <<= 1) { float t = __shfl_up(pfx, o, 64); if (lane >= (uint32_t)o) pfx = __fadd_rn(pfx, t); }
                    const uint32_t adj_k = k + 1 < D ? S.b_adj[k] : NONE;
                    const uint32_t dep_k = k < D ? S.b_dep[k] : 0u;
                    uint32_t i = 0, j = D - 1;
                    float wcur = __shfl(pfx, (int)(D - 1), 64);
                    for (;;) {
                        // Chain hits (elements that are the branching node of everything after them in the range collapse
                        // into the node below, Tree::collapse) are skipped in one go: element k is one iff its depth equals
                        // the minimum adjacent-LCA depth over [k, j), a suffix minimum; the first k that is not decides.
                        bool inr = k >= i && k < j;
                        uint32_t suf = inr ? adj_k : NONE;
#pragma unroll
                        for (int o = 1; o < 64; o <<= 1) {
                            uint32_t t = (uint32_t)__shfl_down((int)suf, o, 64);
                            if (lane + (uint32_t)o < 64) suf = t < suf ? t : suf;
                        }
                        const uint64_t branch = __ballot(inr && dep_k != suf);
                        if (!branch) { ans = S.b_pre[j]; break; }            // a pure chain down to the last element
                        i = (uint32_t)__builtin_ctzll(branch);
                        const uint32_t m = (uint32_t)__shfl((int)suf, (int)i, 64);
                        inr = k >= i && k < j;
                        const uint64_t eq = __ballot(inr && adj_k == m);
                        const uint32_t node = S.b_lca[__builtin_ctzll(eq)];
                        const uint64_t ends = eq | (1ull << j);              // a child ends at every cut and at j
                        const bool is_end = (ends >> k) & 1ull;
                        const uint64_t below = ends & lanemask_lt(k);
                        const uint32_t st = below ? 64u - (uint32_t)__builtin_clzll(below) : i;   // where this child starts
                        const float before_raw = __shfl(pfx, (int)((st - 1u) & 63u), 64);
                        const float seg = is_end ? __fsub_rn(pfx, st ? before_raw : 0.0f) : -1.0f;
                        const float best_w = wave_maxf(seg);
                        const uint32_t best_e = wave_min(is_end && seg == best_w ? k : NONE);   // first maximum, like `wsum > best_w`
                        const uint32_t best_s = (uint32_t)__shfl((int)st, (int)best_e, 64);
                        if (__fdiv_rn(best_w, wcur) < factor) { ans = node; break; }  // src/tree/mix.rs:57-59
                        wcur = best_w;
                        i = best_s;
                        j = best_e;
                    }
                } else if (method == UMGAP_METHOD_TREE) {
                    // ---- tree::mix: greedy descent over the implicit induced tree (uniform code; scored reads, whose
                    // f32 sums depend on the order, and the large-scratch pass) ----
                    uint32_t i = 0, j = D - 1;
                    float wcur = 0.0f;
                    for (uint32_t k = 0; k < D; k++) wcur = __fadd_rn(wcur, S.b_cnt[k]);
                    for (;;) {
                        if (i == j) { ans = S.b_pre[j]; break; }
                        uint32_t m = NONE, node = NONE;
                        for (uint32_t k = i; k < j; k++)
                            if (S.b_adj[k] < m) { m = S.b_adj[k]; node = S.b_lca[k]; }
                        if (S.b_dep[i] == m) { i++; continue; }  // chain hit: collapses into the node below
                        float best_w = -1.0f, wsum = 0.0f;
                        uint32_t best_s = i, best_e = i, s = i;
                        for (uint32_t k = i; k <= j; k++) {
                            wsum = __fadd_rn(wsum, S.b_cnt[k]);
                            if (k == j || S.b_adj[k] == m) {
                                if (wsum > best_w) { best_w = wsum; best_s = s; best_e = k; }
                                s = k + 1;
                                wsum = 0.0f;
                            }
                        }
                        if (__fdiv_rn(best_w, wcur) < factor) { ans = node; break; }  // src/tree/mix.rs:57-59
                        wcur = best_w;
                        i = best_s;
                        j = best_e;
                    }
                } else {
                    // ---- rmq::mix: weights over hits + adjacent LCAs ----
                    const float one_minus = __fsub_rn(1.0f, factor);
                    float best = -__builtin_inff();
                    uint32_t best_p = NONE;
                    const uint32_t n_cand = 2 * D - 1;
                    for (uint32_t cb = 0; cb < n_cand; cb += 64) {
                        uint32_t c = cb + lane;
                        if (c < n_cand) {
                            uint32_t pc = c < D ? S.b_pre[c] : S.b_lca[c - D];
                            uint32_t lc = c < D ? S.b_last[c] : V.last_of_pre[pc];
                            float lw = 0.0f, rw = 0.0f;
                            for (uint32_t k = 0; k < D; k++) {
                                uint32_t pk = S.b_pre[k];
                                if (pc <= pk && pk <= lc) lw = __fadd_rn(lw, S.b_cnt[k]);          // hit below-or-at the candidate
                                if (pk <= pc && pc <= S.b_last[k]) rw = __fadd_rn(rw, S.b_cnt[k]);  // hit above-or-at the candidate
                            }
                            float score = __fadd_rn(__fmul_rn(lw, factor), __fmul_rn(rw, one_minus));  // src/rmq/mix.rs:52-54
                            if (score > best || (score == best && pc < best_p)) { best = score; best_p = pc; }
                        }
                    }
                    float m = wave_maxf(best);
                    ans = wave_min(best == m ? best_p : NONE);
                }
            }
            res_taxon = V.snap_of_pre[ans];  // snapping[aggregate] (src/commands/taxa2agg.rs:178)
            res_status = res_taxon == NONE ? UMGAP_READ_PANIC : UMGAP_READ_OK;
        }
    }
    return true;
}

// the LDS form under its round-1 name
template <bool SCORED, int CAP, int HCAP, bool BOUND = false>
__device__ __forceinline__ bool general_read(const TaxView& V, Scratch<CAP, HCAP>& S, int method, int strategy, float factor, float lower_bound,
                                             const uint32_t* rp, const float* sp, uint64_t n, uint32_t lane, uint32_t& res_taxon, uint32_t& res_status) {
    return general_read_t<SCORED, Scratch<CAP, HCAP>, BOUND>(V, S, method, strategy, factor, lower_bound, rp, sp, n, lane, res_taxon, res_status);
}

}  // namespace agg
}  // namespace umgap
