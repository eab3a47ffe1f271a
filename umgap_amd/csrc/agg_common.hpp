// agg_common.hpp — device helpers shared by the aggregate kernels (gfx950, wave64).
#pragma once
#include "taxonomy.hpp"

namespace umgap {
namespace agg {

constexpr uint32_t NONE = UMGAP_NONE;
constexpr uint64_t NOERR = ~0ull;

// ---- wave reductions: 4 DPP steps inside each row of 16, then 4 readlanes --------------------
template <int CTRL>
__device__ __forceinline__ uint32_t dpp(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xF, 0xF, false);
}
__device__ __forceinline__ uint32_t wave_max(uint32_t v) {  // all lanes must be active
    v = max(v, dpp<0xB1>(v));   // quad_perm [1,0,3,2]
    v = max(v, dpp<0x4E>(v));   // quad_perm [2,3,0,1]
    v = max(v, dpp<0x141>(v));  // row_half_mirror
    v = max(v, dpp<0x140>(v));  // row_mirror
    uint32_t a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    uint32_t c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    return max(max(a, b), max(c, d));
}
__device__ __forceinline__ uint32_t wave_min(uint32_t v) {
    v = min(v, dpp<0xB1>(v));
    v = min(v, dpp<0x4E>(v));
    v = min(v, dpp<0x141>(v));
    v = min(v, dpp<0x140>(v));
    uint32_t a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    uint32_t c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    return min(min(a, b), min(c, d));
}
__device__ __forceinline__ uint64_t wave_min64(uint64_t v) {  // error path only
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, o, 64);
        uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), o, 64);
        uint64_t w = ((uint64_t)hi << 32) | lo;
        v = w < v ? w : v;
    }
    return v;
}

struct Hit {
    uint32_t pre, last;  // pre == NONE: this lane holds no surviving taxon
};

// Position of (one of) the shallowest node(s) with preorder index in [a, b], a <= b, both
// wave-uniform.  All 64 lanes take part: lanes read the partial 64-blocks at both ends (two
// coalesced loads), lanes 0/1 the sparse-table entries covering the whole blocks in between; all
// loads are independent (one round trip), then two DPP reductions.
__device__ __forceinline__ uint32_t wave_argmin_depth(const TaxView& V, uint32_t a, uint32_t b, uint32_t lane) {
    uint32_t d = NONE, pos = NONE;
    const uint32_t ba = a >> 6, bb = b >> 6;
    {   // left partial block (or the whole range when it sits in one block)
        uint32_t p = a + lane, stop = ba == bb ? b : (a | 63u);
        if (p <= stop) { d = V.depth[p]; pos = p; }
    }
    if (bb > ba) {  // right partial block
        uint32_t p = (b & ~63u) + lane;
        if (p <= b) { uint32_t dd = V.depth[p]; if (dd < d) { d = dd; pos = p; } }
        if (bb > ba + 1 && lane < 2) {  // whole blocks ba+1 .. bb-1
            uint32_t len = bb - ba - 1, k = 31u - (uint32_t)__builtin_clz(len);
            uint32_t start = lane == 0 ? ba + 1 : bb - (1u << k);
            uint64_t e = V.sparse[(size_t)k * V.n_blocks + start];
            uint32_t dd = (uint32_t)(e >> 32);
            if (dd < d) { d = dd; pos = (uint32_t)e; }
        }
    }
    const uint32_t m = wave_min(d);
    return wave_min(d == m ? pos : NONE);
}

// Known?  Folds failures into `errkey` (smaller = reported first: BFS level in
// tree::Tree::new (src/tree/mod.rs:35-46), then UnknownTaxon before panic, then payload).
// tree_like: parents[]/ancestors[] walkers (tree::lca, tree::mix, rmq::rtl);
// otherwise rmq::lca / rmq::mix, whose HashMap lookup fails with UnknownTaxon(t) itself.
__device__ __forceinline__ Hit resolve(const TaxView& V, bool tree_like, uint32_t t, uint64_t& errkey) {
    Hit h{NONE, NONE};
    if (t >= V.len) {
        errkey = min(errkey, tree_like ? ((1ull << 32) | t) : (uint64_t)t);
        return h;
    }
    uint2 pl = V.by_id[t];
    if (pl.x == NONE) {
        if (tree_like) {
            uint64_t k = ((uint64_t)V.err_dist[t] << 33) | ((uint64_t)(V.err_kind[t] == 2) << 32) | V.err_payload[t];
            errkey = min(errkey, k);
        } else {
            errkey = min(errkey, (uint64_t)t);  // first_occurence(t) fails (src/rmq/lca.rs:49-54)
        }
        return h;
    }
    h.pre = pl.x;
    h.last = pl.y;
    return h;
}


// wave-level hand-off through LDS: a fence so the compiler keeps the LDS operations on either side in program order,
// and the (free on wave64) wave barrier that re-converges the lanes of a divergent region before the next phase
__device__ __forceinline__ void lds_phase() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Unscored stage 1 through the wave's private LDS hash table (keys[256] all 0 = empty, cnt[256] all
// 0 on entry and on exit).  The three phases -- every lane's ds_add, the leaders' read-back, the key clear -- are
// separated by lds_phase(): the LDS serves one wave's instructions in order, the fence keeps them in this order.
__device__ __forceinline__ void lds_count(uint32_t* keys, uint32_t* cnt, uint32_t t0, uint32_t t1, float lower_bound,
                                          bool& keep0, bool& keep1, float& cnt0, float& cnt1) {
    uint32_t slot0 = 0, slot1 = 0;
    bool lead0 = false, lead1 = false;
    // Neighbouring windows of a protein mostly carry the same taxon: a RUN of equal taxa over consecutive lanes is counted by its
    // first lane alone, with the run's length (round 5).  Sixty-four lanes with one taxon were sixty-four compare-and-swaps and
    // sixty-four adds on ONE LDS address, served one after another -- for reads that follow a proteome that was most of the aggregate
    // stage's time (UMGAP_WALK_DEBUG=16: 4.3 of 10.3 ms).
    const uint32_t p0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)t0, 0x138, 0xF, 0xF, true);      // wave_shr:1 (lane 0 sees 0)
    const uint32_t p1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)t1, 0x138, 0xF, 0xF, true);
    const uint64_t ch0 = __ballot(t0 != p0), ch1 = __ballot(t1 != p1);                                  // where a run starts (runs of misses too)
    const uint32_t lane = __lane_id();
    if (t0 != 0 && t0 != p0) {
        const uint64_t above = (ch0 >> lane) >> 1;
        const uint32_t len = above ? (uint32_t)__builtin_ctzll(above) + 1u : 64u - lane;
        slot0 = (t0 * 0x9E3779B1u) >> 24;
        for (;;) {
            uint32_t old = atomicCAS(&keys[slot0], 0u, t0);
            if (old == 0u) { lead0 = true; break; }
            if (old == t0) break;
            slot0 = (slot0 + 1) & 255u;
        }
        atomicAdd(&cnt[slot0], len);
    }
    if (t1 != 0 && t1 != p1) {
        const uint64_t above = (ch1 >> lane) >> 1;
        const uint32_t len = above ? (uint32_t)__builtin_ctzll(above) + 1u : 64u - lane;
        slot1 = (t1 * 0x9E3779B1u) >> 24;
        for (;;) {
            uint32_t old = atomicCAS(&keys[slot1], 0u, t1);
            if (old == 0u) { lead1 = true; break; }
            if (old == t1) break;
            slot1 = (slot1 + 1) & 255u;
        }
        atomicAdd(&cnt[slot1], len);
    }
    keep0 = keep1 = false;
    cnt0 = cnt1 = 0.0f;
    lds_phase();
    if (lead0) {
        cnt0 = (float)atomicExch(&cnt[slot0], 0u);  // read the multiplicity and reset the slot
        keep0 = cnt0 >= lower_bound;                // agg::filter keeps freq >= lower_bound
    }
    if (lead1) {
        cnt1 = (float)atomicExch(&cnt[slot1], 0u);
        keep1 = cnt1 >= lower_bound;
    }
    // every leader has read its counter before any key is cleared, so no slot is reused early
    lds_phase();
    if (lead0) atomicExch(&keys[slot0], 0u);
    if (lead1) atomicExch(&keys[slot1], 0u);
}


}  // namespace agg
}  // namespace umgap
