// probe_common.hpp — device-side probe primitives of the 9-mer table shared by the lookup kernels
// (index.hip) and the fused classify kernel (classify.hip).  gfx950, wave64.  Layout: common.hpp.
#pragma once
#include "common.hpp"

namespace umgap {
namespace probe {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t quad_xor1(uint32_t v) {  // quad_perm [1,0,3,2]
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);
}
__device__ __forceinline__ uint32_t quad_xor2(uint32_t v) {  // quad_perm [2,3,0,1]
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);
}
__device__ __forceinline__ uint32_t quad_or(uint32_t v) {
    v |= quad_xor1(v);
    v |= quad_xor2(v);
    return v;
}

// Evaluates one 64-byte bucket held by a quad (lane j holds slots 2j, 2j+1 in `e`) against
// `want` = tag | displacement << 28.  Returns bit0 = found, bit1 = bucket overflowed; the taxon of
// the matching slot lands in `val` (quad-uniform).  All four lanes of a quad call this together.
__device__ __forceinline__ uint32_t eval_bucket(u32x4 e, uint32_t want, uint32_t j, uint32_t& val) {
    uint32_t m0 = (e.x & TAG_MASK) == want, m1 = (e.z & TAG_MASK) == want;
    uint32_t v = m0 ? e.y : (m1 ? e.w : 0u);
    uint32_t ovf = (j == 0 && e.x != 0xFFFFFFFFu) ? (e.x >> 31) : 0u;
    val = quad_or(v);
    return quad_or((m0 | m1) | (ovf << 1));
}

// Binary search of the stash (a few thousand entries, L2 resident); reached only by probes that
// exhausted the displacement chain.
__device__ __forceinline__ uint32_t stash_lookup(const TableView& T, uint32_t home, uint32_t tag, uint32_t miss) {
    const uint64_t want = ((uint64_t)home << R_BITS) | tag;
    uint32_t lo = 0, hi = T.n_stash;
    while (lo < hi) {
        uint32_t mid = (lo + hi) >> 1;
        uint64_t k = T.stash_keys[mid];
        if (k == want) return T.stash_vals[mid];
        if (k < want) lo = mid + 1; else hi = mid;
    }
    return miss;
}

// home bucket -> (resident?, local index)
__device__ __forceinline__ bool locate(const TableView& T, uint64_t key, uint32_t& local, uint32_t& tag) {
    uint32_t b;
    probe_of_key(key, T.m, b, tag);
    local = b - T.home_lo;
    return (key >> 45) == 0 && b >= T.home_lo && b < T.home_hi;
}

// The wave-cooperative probe shared by every lookup surface.  Each lane brings one key (already
// turned into local bucket + tag; bit 31 of `tagv` = valid).  The 64 keys are probed in four
// rounds of 16 quad probes: a quad = 4 lanes x 16 B = one 64-byte bucket per load instruction, so
// one wave instruction touches 16 buckets, each as one fully used 64-byte request.  The four
// first-probe loads are issued back to back (4 sectors in flight per quad); a bucket that
// overflowed at build time (rare) sends its quad down the displacement chain afterwards.
// Split in two phases so that a caller can issue several batches of 64 before evaluating any.
struct ProbeBatch {
    uint32_t l[4];      // home bucket (local index) of the round's key, quad-uniform
    uint32_t want[4];   // tag | valid << 31
    u32x4 e[4];         // this lane's 16 bytes of the bucket being looked at
};
__device__ __forceinline__ const u32x4* bucket_ptr(const TableView& T, uint32_t l, uint32_t j) {
    return reinterpret_cast<const u32x4*>(T.slots) + ((size_t)l * 4 + j);
}
__device__ __forceinline__ void probe_issue(const TableView& T, uint32_t local, uint32_t tagv, uint32_t lane, ProbeBatch& B) {
    const uint32_t q = lane >> 2, j = lane & 3;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        int src = 16 * r + (int)q;
        uint32_t l = (uint32_t)__shfl((int)local, src, 64);
        B.want[r] = (uint32_t)__shfl((int)tagv, src, 64);
        B.l[r] = (B.want[r] >> 31) ? l : 0u;  // invalid key: read bucket 0, result discarded
        B.e[r] = __builtin_nontemporal_load(bucket_ptr(T, B.l[r], j));
    }
}
// A batch that was not issued (tail of a group): every key invalid, nothing pending.
__device__ __forceinline__ void probe_skip(ProbeBatch& B) {
#pragma unroll
    for (int r = 0; r < 4; r++) { B.l[r] = 0u; B.want[r] = 0u; B.e[r] = u32x4{0u, 0u, 0u, 0u}; }
}
// res[b][r] (quad-uniform) = taxon of key 16r + (lane >> 2) of batch b, or `miss`.
// The displacement chains of ALL rounds of ALL batches advance together: step d = the loads of bucket home + d
// for every quad that still has to look further (issued back to back), then their evaluation.  Walking the
// chain of each round on its own, as the first version did, serialises up to NB*4 memory round trips per
// call with a handful of requests in flight each -- at load 0.6 that, not the request rate, set the pace.
template <int NB>
__device__ __forceinline__ void probe_eval_multi(const TableView& T, ProbeBatch (&B)[NB], uint32_t miss, uint32_t lane, uint32_t (&res)[NB][4]) {
    const uint32_t j = lane & 3;
    uint32_t found = 0, pend = 0;  // bit b*4+r, quad-uniform: key found / must look in the next bucket
#pragma unroll
    for (int b = 0; b < NB; b++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            uint32_t val;
            uint32_t f = eval_bucket(B[b].e[r], B[b].want[r] & R_MASK, j, val);
            if (!(B[b].want[r] >> 31)) f = 0u;
            res[b][r] = (f & 1u) ? val : miss;
            found |= (f & 1u) << (b * 4 + r);
            pend |= ((f & 3u) == 2u ? 1u : 0u) << (b * 4 + r);
        }
    for (uint32_t d = 1; d <= MAX_DISP && __any(pend != 0u); d++) {
#pragma unroll
        for (int b = 0; b < NB; b++)
#pragma unroll
            for (int r = 0; r < 4; r++)
                if (pend & (1u << (b * 4 + r))) B[b].e[r] = __builtin_nontemporal_load(bucket_ptr(T, B[b].l[r] + d, j));  // halo keeps it in range
        uint32_t next = 0;
#pragma unroll
        for (int b = 0; b < NB; b++)
#pragma unroll
            for (int r = 0; r < 4; r++)
                if (pend & (1u << (b * 4 + r))) {
                    uint32_t val;
                    uint32_t f = eval_bucket(B[b].e[r], (B[b].want[r] & R_MASK) + (d << R_BITS), j, val);
                    if (f & 1u) res[b][r] = val;
                    next |= ((f & 3u) == 2u ? 1u : 0u) << (b * 4 + r);
                }
        pend = next;
    }
    if (pend && T.n_stash) {  // chain exhausted through flagged buckets
#pragma unroll
        for (int b = 0; b < NB; b++)
#pragma unroll
            for (int r = 0; r < 4; r++)
                if (pend & (1u << (b * 4 + r))) res[b][r] = stash_lookup(T, B[b].l[r] + T.home_lo, B[b].want[r] & R_MASK, miss);
    }
    (void)found;
}
__device__ __forceinline__ void probe_eval(const TableView& T, ProbeBatch& B, uint32_t miss, uint32_t lane, uint32_t (&res)[4]) {
    ProbeBatch (&arr)[1] = reinterpret_cast<ProbeBatch (&)[1]>(B);
    uint32_t (&r1)[1][4] = reinterpret_cast<uint32_t (&)[1][4]>(res);
    probe_eval_multi<1>(T, arr, miss, lane, r1);
}
__device__ __forceinline__ void wave_probe64_vals(const TableView& T, uint32_t local, uint32_t tagv, uint32_t miss,
                                                  uint32_t lane, uint32_t (&res)[4]) {
    ProbeBatch B;
    probe_issue(T, local, tagv, lane, B);
    probe_eval(T, B, miss, lane, res);
}

// one key by ONE lane (build passes, scattered look-ups): the chain walked bucket by bucket, the stash behind it
__device__ __forceinline__ uint32_t lane_get(const TableView& T, uint64_t key, uint32_t miss) {
    uint32_t local, tag, res = miss;
    if (!locate(T, key, local, tag)) return miss;
    const uint4* p = reinterpret_cast<const uint4*>(T.slots) + (size_t)local * 4;
    uint32_t want = tag;
    for (uint32_t d = 0;; d++) {
        const uint4 e0 = p[0], e1 = p[1], e2 = p[2], e3 = p[3];
        bool hit = false;
#define UMGAP_CMP(E)                                                  \
    if ((E.x & TAG_MASK) == want) { res = E.y; hit = true; }          \
    if ((E.z & TAG_MASK) == want) { res = E.w; hit = true; }
        UMGAP_CMP(e0) UMGAP_CMP(e1) UMGAP_CMP(e2) UMGAP_CMP(e3)
#undef UMGAP_CMP
        const bool ovf = e0.x != 0xFFFFFFFFu && (e0.x >> 31);
        if (hit || !ovf) break;
        if (d == MAX_DISP) { if (T.n_stash) res = stash_lookup(T, local + T.home_lo, tag, miss); break; }
        p += 4;
        want += 1u << R_BITS;
    }
    return res;
}

}  // namespace probe
}  // namespace umgap
