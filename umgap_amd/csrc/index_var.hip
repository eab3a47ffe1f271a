// index_var.hip — the variable-length (tryptic peptide) index: byte-string key -> taxon, exact.
// Replaces fst::Map::get for keys of 1..63 bytes (src/commands/pept2lca.rs:93 with the tryptic
// index, src/commands/prot2tryp2lca.rs:130; the pipelines use 9..45 residues,
// scripts/umgap-analyse.sh:291).  gfx950 only.
//
// Layout: buckets of 64 B = 4 slots of 16 B, one slot per lane of a quad:
//     w0 = tag (25 bits of the key's hash) | len << 25 (6 bits) | bucket-overflowed << 31 (slot 0 only)
//     w1 = taxon      w2:w3 = offset of the key bytes in the HEAP (all ones = empty slot)
// The heap is the concatenation of the key bytes as given at build time (kept resident).  A probe
// reads the home bucket (hash-chosen); a slot whose tag and length agree is VERIFIED byte by byte
// against the heap, so answers are exact; an overflowed bucket sends the probe on to the next one
// (<= 15 steps, then the stash).  Hits cost two requests (bucket + heap line), misses one.
// INLINE keys (round 2): a key of at most 17 bytes, all in 'A'..'Z' -- most tryptic peptides -- is stored IN its slot at 5 bits per
// residue: bit 63 of the offset word set, its low 63 bits and the 22 low bits of the tag field hold the 85 key bits (the length
// field stays).  Such a slot is compared word for word: exact, and no heap request at all -- one request per lookup, hit or miss.
#include <algorithm>
#include <array>
#include <cmath>
#include <vector>

#include "host_copy.hpp"
#include "var_index.hpp"

using namespace umgap;
using namespace umgap::probe;


namespace {

struct VarFlags { unsigned long long too_full, bad_key, n_stash, mismatch; };
constexpr uint32_t VAR_STASH_CAP = 1u << 20;

__device__ __forceinline__ bool bytes_equal(const uint8_t* a, const uint8_t* b, uint32_t len) {
    if (len >= 8) {
        uint32_t i = 0;
        for (; i + 8 <= len; i += 8)
            if (*reinterpret_cast<const u64_unaligned*>(a + i) != *reinterpret_cast<const u64_unaligned*>(b + i)) return false;
        return i == len || *reinterpret_cast<const u64_unaligned*>(a + len - 8) == *reinterpret_cast<const u64_unaligned*>(b + len - 8);
    }
    for (uint32_t i = 0; i < len; i++)
        if (a[i] != b[i]) return false;
    return true;
}

// does slot word (w0, off) hold the query?  (qin: the query is an inline key (kq, its tag-field bits already in `want`))
__device__ __forceinline__ bool slot_holds(const VarView& T, uint32_t w0, uint64_t off, uint32_t want, bool qin, uint64_t kq, const uint8_t* qp, uint32_t len) {
    if (off == VAR_EMPTY || (w0 & 0x7FFFFFFFu) != want) return false;
    if (qin) return off == kq;
    if (off >> 63) return false;          // an inline slot whose 25 tag-field bits happen to equal the query's tag
    return bytes_equal(T.heap + off, qp, len);
}

__device__ __forceinline__ uint32_t var_stash_lookup(const VarView& T, uint64_t h, const uint8_t* q, uint32_t len, uint32_t miss) {
    uint32_t lo = 0, hi = T.n_stash;
    while (lo < hi) {  // first entry with hash >= h
        uint32_t mid = (lo + hi) >> 1;
        if (T.stash_hash[mid] < h) lo = mid + 1; else hi = mid;
    }
    for (; lo < T.n_stash && T.stash_hash[lo] == h; lo++) {
        uint64_t off = T.stash_off[lo];
        // stash entries keep the length in the top 6 bits of the offset word
        if ((uint32_t)(off >> 58) == len && bytes_equal(T.heap + (off & ((1ull << 58) - 1)), q, len)) return T.stash_val[lo];
    }
    return miss;
}

// One batch of 64 queries (one per lane) straight from global memory: the round-1 kernel body, kept for batches whose bytes
// cannot be staged in LDS (ill-formed offsets: a "key" longer than 63 bytes inside the batch) -- see k_lookup_var below.
__device__ __noinline__ void lookup_batch_global(const uint8_t* __restrict__ qbytes, const uint64_t* __restrict__ qoff, size_t n, size_t base,
                                                    uint32_t miss, uint32_t* __restrict__ out, const VarView T, uint32_t lane, uint32_t min_len, uint32_t max_len) {
    // (T by value, here and in lookup_round_slow: a reference would put the kernel's copy on the stack, and every T.slots / T.nb of
    // the hot loop would become a scratch load whose s_waitcnt vmcnt(0) also waits for whatever the pipelined kernel has in flight)
    const uint32_t q = lane >> 2, j = lane & 3;
    {
        size_t i = base + lane;
        uint64_t h = 0, o = 0;
        uint32_t len = 0, valid = 0;
        if (i < n) {
            o = qoff[i];
            uint64_t l64 = qoff[i + 1] - o;
            if (l64 >= min_len && l64 <= max_len) { len = (uint32_t)l64; h = hash_bytes(qbytes + o, len); valid = 1; }
        }
        uint64_t kq = 0;
        uint32_t khi = 0;
        const uint32_t inl = valid && len <= T.inline_max && pack_inline(qbytes + o, len, kq, khi) ? 1u : 0u;
        const uint32_t wantv = (inl ? khi : ((uint32_t)h & 0x01FFFFFFu)) | (len << 25);
        // Three phases so that the four rounds overlap their memory round trips (walking each round to its end before
        // the next started, as the first version did, kept 16 requests per wave in flight; +6 %.  Fetching all words of
        // a key in one go for hashing / comparison instead of word by word was tried twice (always 8 clamped words; predicated
        // words) and did not pay: the first doubles the cache accesses, both cost registers and with them a resident wave):
        //   1. the home buckets of the 4 rounds (4 loads in flight per quad);
        //   2. the first 8 key bytes in the heap of every slot whose tag and length agree (default cache policy: the
        //      line stays close for phase 3), together with the query's first word;
        //   3. full comparison, then the rare displacement chain / stash.
        uint32_t res[4], v[4], ln[4], want[4], qi[4];
        uint64_t hq[4], bucket[4], kqr[4];
        const uint8_t* qp[4];
        u32x4 e[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            int src = 16 * r + (int)q;
            v[r] = (uint32_t)__shfl((int)valid, src, 64);
            uint32_t hl = (uint32_t)__shfl((int)(uint32_t)h, src, 64), hh = (uint32_t)__shfl((int)(uint32_t)(h >> 32), src, 64);
            uint32_t ol = (uint32_t)__shfl((int)(uint32_t)o, src, 64), oh = (uint32_t)__shfl((int)(uint32_t)(o >> 32), src, 64);
            ln[r] = (uint32_t)__shfl((int)len, src, 64);
            hq[r] = ((uint64_t)hh << 32) | hl;
            qp[r] = qbytes + (((uint64_t)oh << 32) | ol);
            bucket[r] = __umul64hi(hq[r], T.nb);
            want[r] = (uint32_t)__shfl((int)wantv, src, 64);
            qi[r] = (uint32_t)__shfl((int)inl, src, 64);
            kqr[r] = (uint64_t)(uint32_t)__shfl((int)(uint32_t)kq, src, 64) | ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(kq >> 32), src, 64) << 32);
            e[r] = v[r] ? __builtin_nontemporal_load(T.slots + bucket[r] * VAR_SLOTS + j) : u32x4{0u, 0u, 0xFFFFFFFFu, 0xFFFFFFFFu};
        }
        uint64_t hw[4], qw[4];
        bool cand[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const uint64_t off = ((uint64_t)e[r].w << 32) | e[r].z;
            cand[r] = v[r] && off != VAR_EMPTY && (e[r].x & 0x7FFFFFFFu) == want[r] && (qi[r] ? off == kqr[r] : !(off >> 63));
            hw[r] = qw[r] = 0;
            if (cand[r] && !qi[r] && ln[r] >= 8) {
                hw[r] = *reinterpret_cast<const u64_unaligned*>(T.heap + off);
                qw[r] = *reinterpret_cast<const u64_unaligned*>(qp[r]);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            uint32_t r_out = miss;
            if (v[r]) {  // quad-uniform
                const uint64_t off = ((uint64_t)e[r].w << 32) | e[r].z;
                uint32_t val = 0, hit = 0;
                if (cand[r] && (qi[r] || (hw[r] == qw[r] && bytes_equal(T.heap + off, qp[r], ln[r])))) { hit = 1; val = e[r].y; }
                uint32_t ovf = (j == 0) ? (e[r].x >> 31) : 0u;
                uint32_t f = quad_or(hit | (ovf << 1));
                val = quad_or(val);
                bool found = f & 1u, go = !found && (f & 2u);
                if (found) r_out = val;
                for (uint32_t d = 1; go && d <= VAR_MAX_DISP; d++) {   // home bucket overflowed at build time: walk on
                    u32x4 ed = __builtin_nontemporal_load(T.slots + (bucket[r] + d) * VAR_SLOTS + j);
                    uint64_t offd = ((uint64_t)ed.w << 32) | ed.z;
                    val = 0; hit = 0;
                    if (slot_holds(T, ed.x, offd, want[r], qi[r], kqr[r], qp[r], ln[r])) { hit = 1; val = ed.y; }
                    ovf = (j == 0) ? (ed.x >> 31) : 0u;
                    f = quad_or(hit | (ovf << 1));
                    val = quad_or(val);
                    if (f & 1u) { r_out = val; found = true; go = false; }
                    else if (!(f & 2u)) go = false;
                }
                if (!found && go && T.n_stash) r_out = var_stash_lookup(T, hq[r], qp[r], ln[r], miss);  // chain exhausted
            }
            res[r] = r_out;
        }
        if (j == 0) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                size_t k = base + 16 * r + q;
                if (k < n) out[k] = res[r];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The tryptic probe, round 2.  Round 1 read the query bytes from global memory three times per lookup (hash, first word,
// comparison) with per-lane unaligned 8-byte loads -- ~20 vector-cache accesses per lookup, which, not the L2-miss rate, set
// its pace (16.5 G lookups/s at 26 G requests/s).  Now a wave takes two batches of 64 queries at a time:
//   1. the batch's bytes -- one contiguous range of the query buffer, <= 4 KB -- are copied to the wave's LDS strip with
//      aligned 16-byte loads (the only time they are read from global memory);
//   2. every lane hashes its key FROM LDS (aligned dword reads + v_alignbyte);
//   3. the 8 home buckets (2 batches x 4 rounds) are fetched back to back, a quad per bucket;
//   4. for the slot whose tag and length agree the QUAD fetches the key's heap bytes together (lane j: bytes 16j..16j+15,
//      one instruction covers the whole key) and compares them with the query bytes in LDS;
//   5. the rare rest (a second candidate in the bucket, displacement chain, stash) takes the round-1 path.
// ---------------------------------------------------------------------------------------------------------------------
// hash_bytes over staged bytes (same words in the same order)
__device__ __forceinline__ uint64_t hash_lds(const uint32_t* w, uint32_t s, uint32_t len) {
    uint64_t h = 0xCBF29CE484222325ull ^ ((uint64_t)len * 0x9E3779B97F4A7C15ull);
    if (len >= 8) {
        uint32_t i = 0;
        for (; i + 8 <= len; i += 8) {
            h = (h ^ lds_load64(w, s + i)) * 0xD6E8FEB86659FD93ull;
            h ^= h >> 29;
        }
        if (i < len) {
            h = (h ^ lds_load64(w, s + len - 8)) * 0xD6E8FEB86659FD93ull;
            h ^= h >> 29;
        }
    } else {
        for (uint32_t i = 0; i < len; i++) h = (h ^ (lds_load32(w, s + i) & 0xFFu)) * 0x100000001B3ull;
    }
    h ^= h >> 32;
    h *= 0xD6E8FEB86659FD93ull;
    h ^= h >> 32;
    return h;
}

// the rare rest of one round (quad-uniform call): every candidate slot of the home bucket verified by its own lane against the
// query bytes in global memory, then the displacement chain, then the stash -- the round-1 sequence
// (want / qin / kq: what the slot's first word must be, whether the query is an inline key, and its offset word if so -- all
// computed once by the lane that owns the query)
__device__ __noinline__ uint32_t lookup_round_slow(const VarView T, const uint8_t* qp, uint32_t len, uint64_t hq, u32x4 e, uint32_t miss, uint32_t j,
                                                    uint32_t want, bool qin, uint64_t kq) {
    const uint64_t bucket = __umul64hi(hq, T.nb);
    const uint64_t off = ((uint64_t)e.w << 32) | e.z;
    uint32_t hit = 0, vv = 0, r_out = miss;
    if (slot_holds(T, e.x, off, want, qin, kq, qp, len)) { hit = 1; vv = e.y; }
    uint32_t f = quad_or(hit | (((j == 0) ? (e.x >> 31) : 0u) << 1));
    vv = quad_or(vv);
    bool found = f & 1u, go = !found && (f & 2u);
    if (found) r_out = vv;
    for (uint32_t d = 1; go && d <= VAR_MAX_DISP; d++) {
        const u32x4 ed = __builtin_nontemporal_load(T.slots + (bucket + d) * VAR_SLOTS + j);
        const uint64_t offd = ((uint64_t)ed.w << 32) | ed.z;
        vv = 0; hit = 0;
        if (slot_holds(T, ed.x, offd, want, qin, kq, qp, len)) { hit = 1; vv = ed.y; }
        f = quad_or(hit | (((j == 0) ? (ed.x >> 31) : 0u) << 1));
        vv = quad_or(vv);
        if (f & 1u) { r_out = vv; found = true; go = false; }
        else if (!(f & 2u)) go = false;
    }
    if (!found && go && T.n_stash) r_out = var_stash_lookup(T, hq, qp, len, miss);
    return r_out;
}

// ---- steps 2..5 for ONE staged batch (w = the wave's LDS strip that holds its bytes) --------------------------------------------
// per-lane state of a batch: meta = staged offset (13 bits) | len << 13 (6 bits) | valid << 19 | inline << 26; the key of lane l lies
// at a0 + (meta & 0x1FFF) in the query buffer (a0 = the 16-byte aligned absolute address the strip starts at)
struct VarLaneKey {
    uint64_t h, kq;      // hash; offset word of an inline key
    uint32_t meta, wantv;   // wantv: what the slot's first word must be (bit 31 aside)
};

// 2. hash from LDS; short keys packed for the word-for-word comparison with inline slots
__device__ __forceinline__ VarLaneKey var_hash_and_pack(const VarView& T, const uint32_t* w, uint32_t meta) {
    VarLaneKey K{0, 0, meta, 0};
    if ((meta >> 19) & 1u) {
        const uint32_t s = meta & 0x1FFFu, len = (meta >> 13) & 63u;
        K.h = hash_lds(w, s, len);
        uint32_t khi = 0;
        bool inl = false;
        if (len <= T.inline_max) {
            const uint32_t k = s >> 2, sh = s & 3u;
            const uint32_t a0 = w[k], a1 = w[k + 1], a2 = w[k + 2], a3 = w[k + 3], a4 = w[k + 4], a5 = w[k + 5];
            const uint32_t d[5] = {__builtin_amdgcn_alignbyte(a1, a0, sh), __builtin_amdgcn_alignbyte(a2, a1, sh), __builtin_amdgcn_alignbyte(a3, a2, sh),
                                   __builtin_amdgcn_alignbyte(a4, a3, sh), __builtin_amdgcn_alignbyte(a5, a4, sh)};
            inl = pack_inline_words(d, len, K.kq, khi);
        }
        K.wantv = (inl ? khi : ((uint32_t)K.h & 0x01FFFFFFu)) | (len << 25);
        K.meta |= inl ? (1u << 26) : 0u;
    }
    return K;
}

// 3..5: the home buckets of the 4 rounds back to back (a quad per bucket), the candidates' heap bytes (lane j: bytes 16j..16j+15),
// comparison with the query bytes in LDS; the rare rest on the round-1 path
__device__ __forceinline__ void var_probe_staged(const VarView& T, const uint32_t* w, const VarLaneKey& K, uint64_t a0, size_t base, size_t n, uint32_t miss,
                                                 uint32_t* __restrict__ out, uint32_t lane) {
    const uint32_t q = lane >> 2, j = lane & 3;
    uint32_t rm[4], want[4];     // rm: the round's meta (the owning lane's) | first candidate lane << 20 (4 = none) | #candidates << 23
    u32x4 e[4], hp[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int src = 16 * r + (int)q;
        rm[r] = (uint32_t)__shfl((int)K.meta, src, 64);
        const uint32_t hl = (uint32_t)__shfl((int)(uint32_t)K.h, src, 64), hh = (uint32_t)__shfl((int)(uint32_t)(K.h >> 32), src, 64);
        const uint64_t hq = ((uint64_t)hh << 32) | hl;
        want[r] = (uint32_t)__shfl((int)K.wantv, src, 64);
        e[r] = ((rm[r] >> 19) & 1u) ? __builtin_nontemporal_load(T.slots + __umul64hi(hq, T.nb) * VAR_SLOTS + j) : u32x4{0u, 0u, 0xFFFFFFFFu, 0xFFFFFFFFu};
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const uint64_t off = ((uint64_t)e[r].w << 32) | e[r].z;
        const int srcq = 16 * r + (int)q;
        const uint64_t kqr = (uint64_t)(uint32_t)__shfl((int)(uint32_t)K.kq, srcq, 64) | ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(K.kq >> 32), srcq, 64) << 32);
        const bool qin = (rm[r] >> 26) & 1u;
        // an inline query matches its slot word for word (that IS the verification); any other candidate is a heap slot
        const uint32_t cand = (((rm[r] >> 19) & 1u) && off != VAR_EMPTY && (e[r].x & 0x7FFFFFFFu) == want[r] && (qin ? off == kqr : !(off >> 63))) ? 1u : 0u;
        const uint32_t bits = quad_or(cand << j);
        const uint32_t ci = bits ? (uint32_t)__builtin_ctz(bits) : 4u;
        rm[r] |= (ci << 20) | ((uint32_t)__builtin_popcount(bits) << 23);
        const int srcl = (int)((lane & ~3u) | (ci & 3u));
        const uint32_t ol = (uint32_t)__shfl((int)e[r].z, srcl, 64), oh = (uint32_t)__shfl((int)e[r].w, srcl, 64);
        hp[r] = u32x4{0u, 0u, 0u, 0u};
        if (ci < 4u && !qin && 16u * j < ((rm[r] >> 13) & 63u))
            hp[r] = *reinterpret_cast<const u32x4_unaligned*>(T.heap + ((((uint64_t)oh << 32) | ol) + 16u * j));
    }
    uint32_t res[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        uint32_t r_out = miss;
        const uint32_t m = rm[r], ln = (m >> 13) & 63u, ci = (m >> 20) & 7u, nc = (m >> 23) & 7u;
        const bool qin = (m >> 26) & 1u;
        // the key's hash and packed form live in lane 16r + q: fetched while every lane is still active (a shuffle from a lane that
        // the quad-uniform branch below has switched off reads nothing)
        const int src = 16 * r + (int)q;
        const uint64_t hq = (uint64_t)(uint32_t)__shfl((int)(uint32_t)K.h, src, 64) | ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(K.h >> 32), src, 64) << 32);
        const uint64_t kqs = (uint64_t)(uint32_t)__shfl((int)(uint32_t)K.kq, src, 64) | ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(K.kq >> 32), src, 64) << 32);
        if ((m >> 19) & 1u) {   // quad-uniform
            uint32_t mism = 0;
            if (ci < 4u && !qin && 16u * j < ln) {
                const uint32_t s = (m & 0x1FFFu) + 16u * j, k = s >> 2, sh = s & 3u;
                const uint32_t w0 = w[k], w1 = w[k + 1], w2 = w[k + 2], w3 = w[k + 3], w4 = w[k + 4];
                const uint32_t qd[4] = {__builtin_amdgcn_alignbyte(w1, w0, sh), __builtin_amdgcn_alignbyte(w2, w1, sh),
                                        __builtin_amdgcn_alignbyte(w3, w2, sh), __builtin_amdgcn_alignbyte(w4, w3, sh)};
                const uint32_t hd[4] = {hp[r].x, hp[r].y, hp[r].z, hp[r].w};
                const uint32_t left = ln - 16u * j;       // bytes of the key from this lane's piece on (>= 1)
#pragma unroll
                for (uint32_t d = 0; d < 4; d++) {
                    const uint32_t nv = left > 4u * d ? min(left - 4u * d, 4u) : 0u;
                    const uint32_t mask = nv >= 4u ? 0xFFFFFFFFu : ((1u << (8u * nv)) - 1u);
                    mism |= (qd[d] ^ hd[d]) & mask;
                }
            }
            const uint32_t bad = quad_or(mism ? 1u : 0u);
            const uint32_t val = (uint32_t)__shfl((int)e[r].y, (int)((lane & ~3u) | (ci & 3u)), 64);
            const uint32_t ovf = quad_or((j == 0) ? (e[r].x >> 31) : 0u);
            if (ci < 4u && !bad) r_out = val;
            else if (nc > 1u || ovf) {
                // another slot of the bucket shares tag and length (~1e-7), or the home bucket overflowed at build time
                r_out = lookup_round_slow(T, reinterpret_cast<const uint8_t*>(a0 + (m & 0x1FFFu)), ln, hq, e[r], miss, j, want[r], qin, kqs);
            }
        }
        res[r] = r_out;
    }
    if (j == 0) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const size_t k = base + 16 * r + q;
            if (k < n) out[k] = res[r];
        }
    }
}

// The plain kernel: offsets -> staging (aligned 16-byte loads, through registers) -> probe, one batch of 64 queries per wave and step.
__global__ __launch_bounds__(256, 4) void k_lookup_var(const uint8_t* __restrict__ qbytes, const uint64_t* __restrict__ qoff, size_t n,
                                                       uint32_t miss, uint32_t* __restrict__ out, VarView T, uint32_t force_global, uint32_t min_len,
                                                       uint32_t max_len) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[4][VAR_STAGE / 4];
    const uint32_t wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
    const uint64_t all_end = n ? qoff[n] : 0;      // the query buffer is valid on [qbytes, qbytes + qoff[n])
    const uint64_t lo_abs = (uint64_t)(uintptr_t)qbytes, hi_abs = (uint64_t)(uintptr_t)(qbytes + all_end);
    for (size_t base = wave * 64; base < n; base += nwaves * 64) {
        const uint32_t nl = (uint32_t)min((size_t)64, n - base);
        const uint64_t oe = qoff[base + nl];
        const uint64_t o = lane < nl ? qoff[base + lane] : oe, o1 = lane + 1 < nl ? qoff[base + lane + 1] : oe;
        const uint64_t first = (uint64_t)(uint32_t)__shfl((int)(uint32_t)o, 0, 64) | ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(o >> 32), 0, 64) << 32);
        uint32_t meta;
        const VarStage S = var_stage_plan(qbytes, all_end, o, o1, first, oe, force_global, min_len, max_len, meta);
        if (!S.ok) {                                   // ill-formed offsets: the round-1 path (wave-uniform)
            lookup_batch_global(qbytes, qoff, n, base, miss, out, T, lane, min_len, max_len);
            continue;
        }
        u32x4* dst = reinterpret_cast<u32x4*>(lds[wv]);
        for (uint32_t c = lane; (uint64_t)c * 16 < S.span; c += 64) {
            const uint64_t a = S.a0 + (uint64_t)c * 16;
            dst[c] = (a >= lo_abs && a + 16 <= hi_abs) ? __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(a)) : var_edge_chunk(a, lo_abs, hi_abs);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const VarLaneKey K = var_hash_and_pack(T, lds[wv], meta);
        var_probe_staged(T, lds[wv], K, S.a0, base, n, miss, out, lane);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// The pipelined kernel (default).  A batch costs four DEPENDENT round trips in the plain kernel -- offsets, query bytes, bucket,
// heap -- and with 4 waves per SIMD that chain, not the request rate, sets the pace of ordered queries.  Here the first two run
// one and two batches ahead: while batch t probes, the bytes of batch t + 1 travel from global memory straight into the other LDS
// strip (global_load_lds_dwordx4: no registers held, nothing to wait for until the strip is read an iteration later) and the
// offsets of batch t + 2 are in flight into two registers.
__global__ __launch_bounds__(256, 4) void k_lookup_var_pipe(const uint8_t* __restrict__ qbytes, const uint64_t* __restrict__ qoff, size_t n,
                                                            uint32_t miss, uint32_t* __restrict__ out, VarView T, uint32_t min_len, uint32_t max_len) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[4][2][VAR_STAGE / 4];
    const uint32_t wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t step = (((size_t)gridDim.x * blockDim.x) >> 6) * 64;
    size_t base = wave * 64;
    if (base >= n) return;
    const uint64_t all_end = qoff[n];
    const uint64_t lo_abs = (uint64_t)(uintptr_t)qbytes, hi_abs = (uint64_t)(uintptr_t)(qbytes + all_end);
    // offsets of a batch: lane l holds qoff[min(b + l, n)] -- past the end of the list that is the buffer's end, an empty key -- and
    // every lane the batch's end.  Two plain loads whose values nothing looks at before the next iteration (a select between them
    // here would make the wave wait for them, and for the strip transfer issued just before, on the spot).
    auto load_offsets = [&](size_t b, uint64_t& o, uint64_t& oe) {
        o = qoff[min(b + lane, n)];
        oe = qoff[min(b + 64, n)];
    };
    // plans the staging of a batch from its offsets and starts it into strip `buf`; returns the plan, the lane's meta in `meta`
    auto start_stage = [&](uint64_t o, uint64_t oe, uint32_t buf, uint32_t& meta) {
        const uint32_t dl = (uint32_t)__shfl_down((int)(uint32_t)o, 1, 64), dh = (uint32_t)__shfl_down((int)(uint32_t)(o >> 32), 1, 64);
        const uint64_t o1 = lane < 63 ? (((uint64_t)dh << 32) | dl) : oe;
        const uint64_t first = (uint64_t)(uint32_t)__shfl((int)(uint32_t)o, 0, 64) | ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(o >> 32), 0, 64) << 32);
        const VarStage S = var_stage_plan(qbytes, all_end, o, o1, first, oe, 0u, min_len, max_len, meta);
        if (S.ok) {
            uint32_t* strip = lds[wv][buf];
            for (uint32_t c0 = 0; (uint64_t)c0 * 16 < S.span; c0 += 64) {          // wave-uniform trip count (<= 4)
                const uint32_t c = c0 + lane;
                const uint64_t a = S.a0 + (uint64_t)c * 16;
                if ((uint64_t)c * 16 < S.span) {
                    if (a >= lo_abs && a + 16 <= hi_abs)
                        __builtin_amdgcn_global_load_lds(reinterpret_cast<const u32x4*>(a), (__attribute__((address_space(3))) void*)(strip + c0 * 4), 16, 0, 0);
                    else reinterpret_cast<u32x4*>(strip)[c] = var_edge_chunk(a, lo_abs, hi_abs);
                }
            }
        }
        return S;
    };
    uint64_t o_next, oe_next;
    uint32_t meta_cur;
    load_offsets(base, o_next, oe_next);
    VarStage S_cur = start_stage(o_next, oe_next, 0u, meta_cur);
    load_offsets(base + step, o_next, oe_next);
    for (uint32_t t = 0;; t++) {
        const size_t next = base + step, next2 = next + step;
        // everything issued an iteration ago (the strip of this batch, the offsets of the next) has long arrived
        __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const uint32_t* w = lds[wv][t & 1u];
        VarLaneKey K{0, 0, 0, 0};
        if (S_cur.ok) K = var_hash_and_pack(T, w, meta_cur);
        VarStage S_next{0, 0, false};
        uint32_t meta_next = 0;
        if (next < n) S_next = start_stage(o_next, oe_next, (t + 1u) & 1u, meta_next);
        load_offsets(next2, o_next, oe_next);      // unconditional (indexes clamp to n): a value merged under a branch is waited for at the merge
        if (S_cur.ok) var_probe_staged(T, w, K, S_cur.a0, base, n, miss, out, lane);
        else lookup_batch_global(qbytes, qoff, n, base, miss, out, T, lane, min_len, max_len);
        if (next >= n) break;
        base = next;
        S_cur = S_next;
        meta_cur = meta_next;
    }
}

__global__ __launch_bounds__(256) void k_insert_var(u32x4* __restrict__ slots, uint64_t nb, const uint8_t* __restrict__ heap,
                                                    const uint64_t* __restrict__ off, const uint32_t* __restrict__ vals, size_t n,
                                                    VarFlags* flags, uint64_t* __restrict__ stash, uint32_t inline_max) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        uint64_t o = off[i], l64 = off[i + 1] - o;
        if (l64 < 1 || l64 > VAR_MAX_LEN) { atomicAdd(&flags->bad_key, 1ull); continue; }
        uint32_t len = (uint32_t)l64;
        uint64_t h = hash_bytes(heap + o, len);
        uint64_t bucket = __umul64hi(h, nb);
        uint64_t kq = 0;
        uint32_t khi = 0;
        const bool inl = len <= inline_max && pack_inline(heap + o, len, kq, khi);
        const uint32_t w0 = (inl ? khi : ((uint32_t)h & 0x01FFFFFFu)) | (len << 25);
        const unsigned long long word = inl ? kq : o;     // the offset word: the key itself, or where it lies in the heap
        bool placed = false;
        for (uint32_t d = 0; d <= VAR_MAX_DISP && !placed; d++) {
            uint32_t* s = reinterpret_cast<uint32_t*>(slots + (bucket + d) * VAR_SLOTS);
            for (uint32_t k = 0; k < VAR_SLOTS && !placed; k++) {
                unsigned long long* offp = reinterpret_cast<unsigned long long*>(s + 4 * k + 2);  // w2:w3, 8-byte aligned
                if (atomicCAS(offp, (unsigned long long)VAR_EMPTY, word) == VAR_EMPTY) {
                    s[4 * k + 1] = vals[i];
                    atomicOr(s + 4 * k, w0);  // OR, not store: slot 0's word also carries the overflow flag
                    placed = true;
                }
            }
            if (!placed) atomicOr(s, 0x80000000u);  // bucket full: flag it, go on
        }
        if (!placed) {
            unsigned long long at = atomicAdd(&flags->n_stash, 1ull);
            if (at < VAR_STASH_CAP) { stash[3 * at] = h; stash[3 * at + 1] = o | ((uint64_t)len << 58); stash[3 * at + 2] = vals[i]; }
            else atomicAdd(&flags->too_full, 1ull);
        }
    }
}

__global__ void k_fill_var(u32x4* p, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    u32x4 e = {0u, 0u, 0xFFFFFFFFu, 0xFFFFFFFFu};
    for (; i < n; i += stride) p[i] = e;
}

// after the build every key must find ITSELF (its own heap offset's taxon): anything else is a duplicate key
__global__ void k_selfcheck_var(const uint32_t* __restrict__ got, const uint32_t* __restrict__ want, size_t n, VarFlags* flags) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride)
        if (got[i] != want[i]) atomicAdd(&flags->mismatch, 1ull);
}

inline int grid_for(size_t n, size_t per_cu = 8) {
    size_t g = (n + 255) / 256;
    if (g < 1) g = 1;
    if (g > 256 * per_cu) g = 256 * per_cu;
    return (int)g;
}

}  // namespace

namespace umgap {

bool load_file_range(int fd, off_t file_off, size_t total, void* d_dst, int device);   // index.hip

// the packed format (index_varp.hip)
int varp_scan(const uint8_t* d_bytes, const uint64_t* d_off, size_t n, hipStream_t stream, uint64_t* bad_len, uint64_t* not_az, uint64_t* slots);
int varp_build(umgap_var_index* v, const uint8_t* d_bytes, const uint64_t* d_off, const uint32_t* d_vals, size_t n, uint64_t total_slots, double load,
               hipStream_t stream);
int varp_lookup_dev(const umgap_var_index* v, const uint8_t* d_bytes, const uint64_t* d_off, size_t n, uint32_t miss, uint32_t* d_out, hipStream_t s,
                    uint32_t min_len, uint32_t max_len);
int varp_dump(const umgap_var_index* v, uint8_t* heap_out, uint64_t heap_cap, uint64_t* offsets_out, uint32_t* lens_out, uint32_t* taxa_out, size_t cap,
              size_t* n_out, uint64_t* heap_bytes_out);

void var_free(umgap_var_index* v) {
    if (!v) return;
    void* ptrs[] = {v->d_slots, v->d_heap, v->d_stash_hash, v->d_stash_off, v->d_stash_val, v->d_pstash};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    delete v;
}

uint64_t var_table_bytes(const umgap_var_index* v) { return v ? (uint64_t)(v->nb + VAR_MAX_DISP) * 64 + v->heap_bytes : 0; }

int var_set_stash(umgap_var_index* v, std::vector<std::array<uint64_t, 3>>& e) {
    if (e.empty()) return UMGAP_OK;
    std::sort(e.begin(), e.end());
    std::vector<uint64_t> h(e.size()), o(e.size());
    std::vector<uint32_t> t(e.size());
    for (size_t i = 0; i < e.size(); i++) { h[i] = e[i][0]; o[i] = e[i][1]; t[i] = (uint32_t)e[i][2]; }
    UMGAP_HIP(hipMalloc((void**)&v->d_stash_hash, h.size() * 8));
    UMGAP_HIP(hipMalloc((void**)&v->d_stash_off, o.size() * 8));
    UMGAP_HIP(hipMalloc((void**)&v->d_stash_val, t.size() * 4));
    UMGAP_HIP(hipMemcpy(v->d_stash_hash, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    UMGAP_HIP(hipMemcpy(v->d_stash_off, o.data(), o.size() * 8, hipMemcpyHostToDevice));
    UMGAP_HIP(hipMemcpy(v->d_stash_val, t.data(), t.size() * 4, hipMemcpyHostToDevice));
    v->n_stash = (uint32_t)e.size();
    return UMGAP_OK;
}

// UMGAP_VAR_GLOBAL=1: every batch takes the round-1 global-memory path (A/B measurements, debugging)
static uint32_t var_force_global() {
    const char* e = getenv("UMGAP_VAR_GLOBAL");
    return (e && e[0] == '1') ? 1u : 0u;
}

// d_heap / d_off / d_vals: device arrays; the heap is adopted by the index (not copied) when own_heap.
int var_build(umgap_var_index** out, uint8_t* d_heap, uint64_t heap_bytes, const uint64_t* d_off, const uint32_t* d_vals, size_t n,
              double load, hipStream_t stream) {
    if (!(load > 0.0 && load < 1.0)) load = 0.5;
    auto* v = new umgap_var_index();
    v->d_heap = d_heap;
    v->heap_bytes = heap_bytes;
    v->n_keys = n;
    {
        const char* e = getenv("UMGAP_VAR_INLINE");      // A/B switch of the measurement tools: 0 = every key in the heap
        v->inline_max = (e && e[0] == '0') ? 0u : VAR_INLINE_MAX;
    }
    VarFlags* d_flags = nullptr;
    uint64_t* d_stash = nullptr;
    uint32_t* d_got = nullptr;
    int rc = UMGAP_OK;
    auto cleanup = [&]() { if (d_flags) (void)hipFree(d_flags); if (d_stash) (void)hipFree(d_stash); if (d_got) (void)hipFree(d_got); };
    if (hipMalloc((void**)&d_flags, sizeof(VarFlags)) != hipSuccess || hipMalloc((void**)&d_stash, (size_t)VAR_STASH_CAP * 24) != hipSuccess ||
        hipMalloc((void**)&d_got, std::max<size_t>(n, 1) * 4) != hipSuccess) { cleanup(); v->d_heap = nullptr; var_free(v); return fail(UMGAP_ERR_NOMEM, "hipMalloc(var index build)"); }
    uint64_t nb = std::max<uint64_t>(1024, (uint64_t)std::ceil((double)n / (VAR_SLOTS * load)));
    for (int attempt = 0; attempt < 4; attempt++) {
        if (v->d_slots) { (void)hipFree(v->d_slots); v->d_slots = nullptr; }
        size_t nslots = (size_t)(nb + VAR_MAX_DISP) * VAR_SLOTS;
        if (hipMalloc((void**)&v->d_slots, nslots * 16) != hipSuccess) { rc = fail(UMGAP_ERR_NOMEM, "hipMalloc of the var table failed"); break; }
        v->nb = nb;
        hipLaunchKernelGGL(k_fill_var, dim3(grid_for(nslots)), dim3(256), 0, stream, v->d_slots, nslots);
        (void)hipMemsetAsync(d_flags, 0, sizeof(VarFlags), stream);
        if (n) hipLaunchKernelGGL(k_insert_var, dim3(grid_for(n)), dim3(256), 0, stream, v->d_slots, nb, d_heap, d_off, d_vals, n, d_flags, d_stash, v->inline_max);
        VarFlags h{};
        hipError_t e = hipMemcpyAsync(&h, d_flags, sizeof(h), hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        if (e != hipSuccess) { rc = fail(UMGAP_ERR_DEVICE, std::string("var index build: ") + hipGetErrorString(e)); break; }
        if (h.bad_key) { rc = fail(UMGAP_ERR_BUILD, "index build: " + std::to_string(h.bad_key) + " keys are empty or longer than 63 bytes"); break; }
        if (h.too_full) { nb = nb + nb / 4 + 1; rc = fail(UMGAP_ERR_BUILD, "index build: table too full"); continue; }
        std::vector<uint64_t> raw(3 * (size_t)h.n_stash);
        if (h.n_stash && hipMemcpy(raw.data(), d_stash, raw.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(UMGAP_ERR_DEVICE, "stash copy failed"); break; }
        std::vector<std::array<uint64_t, 3>> entries(h.n_stash);
        for (size_t i = 0; i < h.n_stash; i++) entries[i] = {raw[3 * i], raw[3 * i + 1], raw[3 * i + 2]};
        rc = var_set_stash(v, entries);
        if (rc) break;
        // self check: every key finds its own taxon (a duplicate key finds the other copy's)
        if (n) {
            hipLaunchKernelGGL(k_lookup_var, dim3(grid_for(n)), dim3(256), 0, stream, d_heap, d_off, n, 0xFFFFFFFFu, d_got, v->view(), var_force_global(), 1u, VAR_MAX_LEN);
            hipLaunchKernelGGL(k_selfcheck_var, dim3(grid_for(n)), dim3(256), 0, stream, d_got, d_vals, n, d_flags);
            e = hipMemcpyAsync(&h, d_flags, sizeof(h), hipMemcpyDeviceToHost, stream);
            if (e == hipSuccess) e = hipStreamSynchronize(stream);
            if (e != hipSuccess) { rc = fail(UMGAP_ERR_DEVICE, std::string("var index self-check: ") + hipGetErrorString(e)); break; }
            if (h.mismatch) { rc = fail(UMGAP_ERR_BUILD, "index build: duplicate keys (keys must be distinct, cf. fst::MapBuilder::insert)"); break; }
        }
        rc = UMGAP_OK;
        break;
    }
    cleanup();
    if (rc) { v->d_heap = nullptr; var_free(v); return rc; }
    *out = v;
    return UMGAP_OK;
}

int var_lookup_dev(const umgap_var_index* v, const uint8_t* d_bytes, const uint64_t* d_off, size_t n, uint32_t miss, uint32_t* d_out, hipStream_t s,
                   uint32_t min_len, uint32_t max_len) {
    if (min_len < 1) min_len = 1;
    if (max_len > VAR_MAX_LEN) max_len = VAR_MAX_LEN;
    if (v->packed()) return varp_lookup_dev(v, d_bytes, d_off, n, miss, d_out, s, min_len, max_len);
    // a grid of whole resident rounds (two of them): with a partial last round the kernel measured 16 % slower
    static int per_cu[2] = {0, 0};
    static int pipe = -1;
    if (pipe < 0) {
        const char* env = getenv("UMGAP_VAR_PIPE");       // A/B switch of the measurement tools: 0 = the plain kernel
        pipe = (env && env[0] == '0') ? 0 : 1;
    }
    const int which = (pipe && !var_force_global()) ? 1 : 0;
    if (!per_cu[which]) {
        int b = 0;
        hipError_t e = which ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, k_lookup_var_pipe, 256, 0) : hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, k_lookup_var, 256, 0);
        if (e != hipSuccess || b < 1) b = 4;
        per_cu[which] = 2 * b;
    }
    if (n) {
        if (which) hipLaunchKernelGGL(k_lookup_var_pipe, dim3(grid_for(n, (size_t)per_cu[which])), dim3(256), 0, s, d_bytes, d_off, n, miss, d_out, v->view(), min_len, max_len);
        else hipLaunchKernelGGL(k_lookup_var, dim3(grid_for(n, (size_t)per_cu[which])), dim3(256), 0, s, d_bytes, d_off, n, miss, d_out, v->view(), var_force_global(), min_len, max_len);
    }
    UMGAP_HIP(hipGetLastError());
    return UMGAP_OK;
}

}  // namespace umgap

// ==========================================================================================
// C ABI (variable-length keys)
// ==========================================================================================
namespace umgap { int use_device(int device); }

namespace {
int new_var_handle(int device, umgap_index** out) {
    int rc = use_device(device);
    if (rc) return rc;
    umgap_index* ix = new umgap_index();
    ix->device = device;
    ix->kind = 2;
    hipError_t e = hipStreamCreateWithFlags(&ix->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete ix; return fail(UMGAP_ERR_DEVICE, std::string("hipStreamCreate: ") + hipGetErrorString(e)); }
    // a build reads the caller's device arrays on the handle's own (non-blocking) stream: whatever produced them, on whichever
    // stream, must have finished -- once per handle, before a build that takes seconds
    e = hipDeviceSynchronize();
    if (e != hipSuccess) { (void)hipStreamDestroy(ix->stream); delete ix; return fail(UMGAP_ERR_DEVICE, std::string("hipDeviceSynchronize: ") + hipGetErrorString(e)); }
    *out = ix;
    return UMGAP_OK;
}
}  // namespace

extern "C" {

int umgap_index_kind(const umgap_index* ix) { return ix ? ix->kind : 0; }
int umgap_index_var_layout(const umgap_index* ix) { return (ix && ix->kind == 2 && ix->var) ? (ix->var->packed() ? 1 : 0) : -1; }

int umgap_index_build_var_dev(const uint8_t* d_bytes, const uint64_t* d_offsets, const uint32_t* d_taxa, size_t n, uint64_t n_bytes,
                              double load_factor, int device, umgap_index** out) {
    if (!out || (n && (!d_bytes || !d_offsets || !d_taxa))) return fail(UMGAP_ERR_INVALID, "umgap_index_build_var_dev: null argument");
    umgap_index* ix = nullptr;
    int rc = new_var_handle(device, &ix);
    if (rc) return rc;
    // Words over 'A'..'Z' (every tryptic index) take the packed format: the keys live in the table, the bytes are only read here.
    // UMGAP_VAR_PACKED=0 keeps the legacy format (slots + heap) for them too (A/B measurements, tests of that path).
    {
        const char* env = getenv("UMGAP_VAR_PACKED");
        uint64_t bad_len = 0, not_az = 0, slots = 0;
        if (!(env && env[0] == '0') && n) {
            rc = varp_scan(d_bytes, d_offsets, n, ix->stream, &bad_len, &not_az, &slots);
            if (rc) { umgap_index_free(ix); return rc; }
            if (bad_len) { umgap_index_free(ix); return fail(UMGAP_ERR_BUILD, "index build: " + std::to_string(bad_len) + " keys are empty or longer than 63 bytes"); }
            if (!not_az) {
                ix->var = new umgap_var_index();
                rc = varp_build(ix->var, d_bytes, d_offsets, d_taxa, n, slots, load_factor, ix->stream);
                if (rc) { umgap_index_free(ix); return rc; }
                ix->n_keys = n;
                *out = ix;
                return UMGAP_OK;
            }
        }
    }
    uint8_t* heap = nullptr;
    hipError_t e = hipMalloc((void**)&heap, n_bytes + 64);   // the probe's quad reads 16-byte pieces: up to 15 bytes past a key's end
    if (e == hipSuccess && n_bytes) e = hipMemcpyAsync(heap, d_bytes, n_bytes, hipMemcpyDeviceToDevice, ix->stream);
    if (e != hipSuccess) { if (heap) (void)hipFree(heap); umgap_index_free(ix); return fail(UMGAP_ERR_NOMEM, std::string("var index heap: ") + hipGetErrorString(e)); }
    rc = var_build(&ix->var, heap, n_bytes, d_offsets, d_taxa, n, load_factor, ix->stream);
    if (rc) { (void)hipFree(heap); umgap_index_free(ix); return rc; }
    ix->n_keys = n;
    *out = ix;
    return UMGAP_OK;
}

int umgap_index_build_var(const uint8_t* bytes, const uint64_t* offsets, const uint32_t* taxa, size_t n, double load_factor, int device,
                          umgap_index** out) {
    if (!out || (n && (!bytes || !offsets || !taxa))) return fail(UMGAP_ERR_INVALID, "umgap_index_build_var: null argument");
    int rc = use_device(device);
    if (rc) return rc;
    const uint64_t base = n ? offsets[0] : 0, n_bytes = n ? offsets[n] - base : 0;
    std::vector<uint64_t> rel(n + 1, 0);
    for (size_t i = 0; i <= n && n; i++) rel[i] = offsets[i] - base;
    uint8_t* db = nullptr; uint64_t* doff = nullptr; uint32_t* dv = nullptr;
    hipError_t e = hipMalloc((void**)&db, n_bytes + 64);   // becomes the heap: padded for the probe's 16-byte pieces
    if (e == hipSuccess) e = hipMalloc((void**)&doff, (n + 1) * 8);
    if (e == hipSuccess) e = hipMalloc((void**)&dv, std::max<size_t>(n, 1) * 4);
    if (e == hipSuccess && n_bytes) e = hipMemcpy(db, bytes + base, n_bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(doff, rel.data(), (n + 1) * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess && n) e = hipMemcpy(dv, taxa, n * 4, hipMemcpyHostToDevice);
    if (e != hipSuccess) rc = fail(UMGAP_ERR_NOMEM, std::string("umgap_index_build_var: ") + hipGetErrorString(e));
    else rc = umgap_index_build_var_dev(db, doff, dv, n, n_bytes, load_factor, device, out);
    if (db) (void)hipFree(db);
    if (doff) (void)hipFree(doff);
    if (dv) (void)hipFree(dv);
    return rc;
}

// dump (printindex): offsets_out[i] / lens_out[i] locate key i in the heap copy `heap_out` (heap_cap bytes)
int umgap_index_dump_var(const umgap_index* ix, uint8_t* heap_out, uint64_t heap_cap, uint64_t* offsets_out, uint32_t* lens_out,
                         uint32_t* taxa_out, size_t cap, size_t* n_out, uint64_t* heap_bytes_out) {
    if (!ix || ix->kind != 2 || !ix->var || !n_out || !heap_bytes_out) return fail(UMGAP_ERR_INVALID, "umgap_index_dump_var: not a variable-length index");
    if (ix->bound_to) return fail(UMGAP_ERR_INVALID, "the index is bound to a taxonomy (umgap_index_bind_taxonomy): its values are interval codes, not taxon ids; unbind it first");
    int rc = use_device(ix->device);
    if (rc) return rc;
    const umgap_var_index* v = ix->var;
    if (v->packed()) return varp_dump(v, heap_out, heap_cap, offsets_out, lens_out, taxa_out, cap, n_out, heap_bytes_out);
    *heap_bytes_out = v->heap_bytes;
    if (heap_out && heap_cap >= v->heap_bytes && v->heap_bytes) UMGAP_HIP(hipMemcpy(heap_out, v->d_heap, v->heap_bytes, hipMemcpyDeviceToHost));
    size_t n = 0;
    uint64_t appended = 0;      // inline keys are unpacked behind the heap copy
    const size_t total = (size_t)(v->nb + VAR_MAX_DISP) * VAR_SLOTS, chunk = 1u << 22;
    std::vector<uint32_t> buf(chunk * 4);
    for (size_t s0 = 0; s0 < total; s0 += chunk) {
        size_t cnt = std::min(chunk, total - s0);
        UMGAP_HIP(hipMemcpy(buf.data(), v->d_slots + s0, cnt * 16, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < cnt; i++) {
            uint64_t off = ((uint64_t)buf[4 * i + 3] << 32) | buf[4 * i + 2];
            if (off == VAR_EMPTY) continue;
            const uint32_t len = (buf[4 * i] >> 25) & 63u;
            if (off & VAR_INLINE) {
                uint8_t key[VAR_INLINE_MAX];
                unpack_inline(buf[4 * i], off, key);
                off = v->heap_bytes + appended;
                if (heap_out && heap_cap >= off + len) memcpy(heap_out + off, key, len);
                appended += len;
            }
            if (n < cap) { offsets_out[n] = off; lens_out[n] = len; taxa_out[n] = buf[4 * i + 1]; }
            n++;
        }
    }
    *heap_bytes_out = v->heap_bytes + appended;
    if (v->n_stash) {
        std::vector<uint64_t> so(v->n_stash);
        std::vector<uint32_t> sv(v->n_stash);
        UMGAP_HIP(hipMemcpy(so.data(), v->d_stash_off, so.size() * 8, hipMemcpyDeviceToHost));
        UMGAP_HIP(hipMemcpy(sv.data(), v->d_stash_val, sv.size() * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < so.size(); i++) {
            if (n < cap) { offsets_out[n] = so[i] & ((1ull << 58) - 1); lens_out[n] = (uint32_t)(so[i] >> 58); taxa_out[n] = sv[i]; }
            n++;
        }
    }
    *n_out = n;
    return UMGAP_OK;
}

int umgap_index_lookup_var_dev(umgap_index* ix, const uint8_t* d_bytes, const uint64_t* d_offsets, size_t n, uint32_t miss_value,
                               uint32_t* d_out, void* stream) {
    if (!ix || ix->kind != 2 || !ix->var) return fail(UMGAP_ERR_INVALID, "umgap_index_lookup_var_dev: not a variable-length index");
    if (n && (!d_bytes || !d_offsets || !d_out)) return fail(UMGAP_ERR_INVALID, "umgap_index_lookup_var_dev: null argument");
    if (ix->bound_to) return fail(UMGAP_ERR_INVALID, "the index is bound to a taxonomy (umgap_index_bind_taxonomy): its values are interval codes, not taxon ids; unbind it first");
    int rc = use_device(ix->device);
    if (rc) return rc;
    return var_lookup_dev(ix->var, d_bytes, d_offsets, n, miss_value, d_out, (hipStream_t)stream, 1u, VAR_MAX_LEN);
}

}  // extern "C"
namespace umgap { int ensure_stage(umgap_index* ix); }   // index.hip
namespace {
__global__ __launch_bounds__(256) void k_rebase_offsets(uint64_t* __restrict__ off, size_t n, uint64_t base) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += step) off[i] -= base;
}
}  // namespace
extern "C" {

// Host arrays in, host array out, the link included: chunks of at most 512 K peptides / 17 MB of residues through the handle's three
// staging slots -- an upload lane and a probe lane tied by events, as umgap_index_lookup9 (index.hip); arrays in pinned memory
// are copied from / to directly, pageable ones through the pinned staging buffers with a host memcpy.  (Until late round 3: three
// hipMallocs, one upload of everything, one launch, one download and three hipFrees per call: 0.32 G look-ups/s.)
int umgap_index_lookup_var(umgap_index* ix, const uint8_t* bytes, const uint64_t* offsets, size_t n, uint32_t miss_value, uint32_t* out) {
    if (!ix || ix->kind != 2 || !ix->var) return fail(UMGAP_ERR_INVALID, "umgap_index_lookup_var: not a variable-length index");
    if (n && (!bytes || !offsets || !out)) return fail(UMGAP_ERR_INVALID, "umgap_index_lookup_var: null argument");
    if (ix->bound_to) return fail(UMGAP_ERR_INVALID, "the index is bound to a taxonomy (umgap_index_bind_taxonomy): its values are interval codes, not taxon ids; unbind it first");
    int rc = use_device(ix->device);
    if (rc) return rc;
    if (!n) return UMGAP_OK;
    rc = ensure_stage(ix);
    if (rc) return rc;
    constexpr int NS = 3;
    constexpr size_t CAPN = 1u << 19, CAPB = 17u << 20, PAD = 256;     // a slot (24 MB): [pad | bytes | pad][offsets][answers]
    static_assert(PAD + CAPB + PAD + (CAPN + 1) * 8 + CAPN * 4 <= (size_t)(2u << 20) * 12, "a chunk must fit a staging slot");
    auto pinned = [](const void* p) {
        hipPointerAttribute_t a{};
        if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
        return a.type == hipMemoryTypeHost;
    };
    const bool in_pinned = pinned(bytes + offsets[0]) && pinned(bytes + offsets[n] - (offsets[n] > offsets[0] ? 1 : 0)) && pinned(offsets) && pinned(offsets + n);
    const bool out_pinned = pinned(out) && pinned(out + n - 1);
    struct Events {
        hipEvent_t uploaded[NS] = {}, done[NS] = {};
        ~Events() { for (hipEvent_t x : uploaded) if (x) (void)hipEventDestroy(x); for (hipEvent_t x : done) if (x) (void)hipEventDestroy(x); }
    } ev;
    for (int i = 0; i < NS; i++) {
        UMGAP_HIP(hipEventCreateWithFlags(&ev.uploaded[i], hipEventDisableTiming));
        UMGAP_HIP(hipEventCreateWithFlags(&ev.done[i], hipEventDisableTiming));
    }
    const hipStream_t up = ix->up_stream, run = ix->run_stream;
    size_t pending_at[NS] = {}, pending_len[NS] = {};
    auto h_bytes = [&](int s) { return (uint8_t*)ix->h_stage[s] + PAD; };
    auto h_off = [&](int s) { return (uint64_t*)((char*)ix->h_stage[s] + PAD + CAPB + PAD); };
    auto h_out = [&](int s) { return (uint32_t*)((char*)ix->h_stage[s] + PAD + CAPB + PAD + (CAPN + 1) * 8); };
    auto d_bytes = [&](int s) { return (uint8_t*)ix->d_stage[s] + PAD; };
    auto d_off = [&](int s) { return (uint64_t*)((char*)ix->d_stage[s] + PAD + CAPB + PAD); };
    auto d_out = [&](int s) { return (uint32_t*)((char*)ix->d_stage[s] + PAD + CAPB + PAD + (CAPN + 1) * 8); };
    auto drain = [&](int s) -> int {
        if (!pending_len[s]) return UMGAP_OK;
        UMGAP_HIP(hipEventSynchronize(ev.done[s]));
        if (!out_pinned) par_memcpy(out + pending_at[s], h_out(s), pending_len[s] * 4);
        pending_len[s] = 0;
        return UMGAP_OK;
    };
    size_t c = 0;
    for (size_t i0 = 0; i0 < n; c++) {
        // the chunk: as many whole peptides as fit (a peptide beyond VAR_MAX_LEN cannot be a key; one beyond CAPB is refused)
        size_t lo = i0 + 1, hi = std::min(n, i0 + CAPN);
        const uint64_t b0 = offsets[i0];
        if (offsets[i0 + 1] < b0 || offsets[i0 + 1] - b0 > CAPB) { rc = fail(UMGAP_ERR_INVALID, "umgap_index_lookup_var: offsets must not decrease, and a single query is limited to 17 MB"); break; }
        while (lo < hi) {
            const size_t mid = lo + (hi - lo + 1) / 2;
            if (offsets[mid] >= b0 && offsets[mid] - b0 <= CAPB) lo = mid; else hi = mid - 1;
        }
        const size_t i1 = lo, len = i1 - i0;
        const uint64_t nb = offsets[i1] - b0;
        const int slot = (int)(c % NS);
        rc = drain(slot);
        if (rc) break;
        const uint8_t* src_b = bytes + b0;
        const uint64_t* src_o = offsets + i0;
        if (!in_pinned) {
            par_memcpy(h_bytes(slot), src_b, nb);
            par_memcpy(h_off(slot), src_o, (len + 1) * 8);
            src_b = h_bytes(slot);
            src_o = h_off(slot);
        }
        if (nb) UMGAP_HIP_BREAK(hipMemcpyAsync(d_bytes(slot), src_b, nb, hipMemcpyHostToDevice, up));
        UMGAP_HIP_BREAK(hipMemcpyAsync(d_off(slot), src_o, (len + 1) * 8, hipMemcpyHostToDevice, up));
        UMGAP_HIP_BREAK(hipEventRecord(ev.uploaded[slot], up));
        UMGAP_HIP_BREAK(hipStreamWaitEvent(run, ev.uploaded[slot], 0));
        hipLaunchKernelGGL(k_rebase_offsets, dim3((unsigned)std::min<size_t>((len + 256) / 256, 1024)), dim3(256), 0, run, d_off(slot), len + 1, b0);
        rc = var_lookup_dev(ix->var, d_bytes(slot), d_off(slot), len, miss_value, d_out(slot), run, 1u, VAR_MAX_LEN);
        if (rc) break;
        UMGAP_HIP_BREAK(hipMemcpyAsync(out_pinned ? (void*)(out + i0) : (void*)h_out(slot), d_out(slot), len * 4, hipMemcpyDeviceToHost, run));
        UMGAP_HIP_BREAK(hipEventRecord(ev.done[slot], run));
        pending_at[slot] = i0;
        pending_len[slot] = len;
        i0 = i1;
    }
    for (int i = 0; i < NS; i++) { const int r2 = drain(i); if (!rc) rc = r2; }
    if (rc) { (void)hipStreamSynchronize(up); (void)hipStreamSynchronize(run); }
    return rc;
}

}  // extern "C"

namespace umgap {

// "UMGAPIX1" kind 2: header, slots, heap, stash {hash, off|len<<58, taxon}
int var_save(const umgap_index* ix, FILE* f) {
    const umgap_var_index* v = ix->var;
    IndexFileHeader h{};
    memcpy(h.magic, "UMGAPIX1", 8);
    h.version = 1; h.kind = 2; h.n_keys = v->n_keys; h.m = v->inline_max; h.nb = 0;
    h.mix_c1 = v->nb; h.mix_c2 = v->heap_bytes; h.r_bits = VAR_MAX_DISP; h.slots = VAR_SLOTS; h.n_stash = v->n_stash;
    bool ok = fwrite(&h, sizeof(h), 1, f) == 1;
    const size_t chunk = 64u << 20;
    std::vector<char> buf(chunk);
    auto dump = [&](const void* d, size_t total) {
        for (size_t off = 0; ok && off < total; off += chunk) {
            size_t len = std::min(chunk, total - off);
            if (hipMemcpy(buf.data(), (const char*)d + off, len, hipMemcpyDeviceToHost) != hipSuccess) { ok = false; break; }
            ok = fwrite(buf.data(), 1, len, f) == len;
        }
    };
    dump(v->d_slots, (size_t)(v->nb + VAR_MAX_DISP) * 64);
    if (v->packed()) dump(v->d_pstash, (size_t)v->n_stash * sizeof(VarpStashEntry));     // no heap: the keys are in the slots
    else {
        dump(v->d_heap, v->heap_bytes);
        dump(v->d_stash_hash, (size_t)v->n_stash * 8);
        dump(v->d_stash_off, (size_t)v->n_stash * 8);
        dump(v->d_stash_val, (size_t)v->n_stash * 4);
    }
    return ok ? UMGAP_OK : fail(UMGAP_ERR_IO, "error writing the index file");
}

int var_open(umgap_index* ix, const IndexFileHeader& h, FILE* f, const char* path) {
    if (h.slots != VAR_SLOTS || h.r_bits != VAR_MAX_DISP) return fail(UMGAP_ERR_FORMAT, std::string(path) + ": unsupported UMGAPIX1 (var) parameters");
    auto* v = new umgap_var_index();
    ix->var = v;
    ix->kind = 2;
    v->nb = h.mix_c1; v->heap_bytes = h.mix_c2; v->n_keys = h.n_keys; v->n_stash = (uint32_t)h.n_stash;
    if (h.m > VAR_INLINE_MAX && h.m != VARP_INLINE_ALL) return fail(UMGAP_ERR_FORMAT, std::string(path) + ": unsupported UMGAPIX1 (var) inline-key length");
    v->inline_max = h.m;      // 0 in files written before inline keys existed: every slot points into the heap
    if (v->packed() && v->nb >= (1ull << 32)) return fail(UMGAP_ERR_FORMAT, std::string(path) + ": packed var table with 2^32 buckets or more");
    ix->n_keys = h.n_keys;
    // the sections follow the header back to back; each is pulled in by the parallel loader of the 9-mer tables (index.hip)
    bool ok = true;
    const int fd = fileno(f);
    off_t at = (off_t)sizeof(IndexFileHeader);
    auto slurp = [&](void** d, size_t total) {
        if (!ok) return;
        if (hipMalloc(d, total + 64) != hipSuccess) { ok = false; return; }   // + 64: the heap is read in 16-byte pieces past a key's end
        ok = load_file_range(fd, at, total, *d, ix->device);
        at += (off_t)total;
    };
    slurp((void**)&v->d_slots, (size_t)(v->nb + VAR_MAX_DISP) * 64);
    if (v->packed()) {
        if (v->n_stash) slurp((void**)&v->d_pstash, (size_t)v->n_stash * sizeof(VarpStashEntry));
    } else {
        slurp((void**)&v->d_heap, v->heap_bytes);
        slurp((void**)&v->d_stash_hash, (size_t)v->n_stash * 8);
        slurp((void**)&v->d_stash_off, (size_t)v->n_stash * 8);
        slurp((void**)&v->d_stash_val, (size_t)v->n_stash * 4);
    }
    if (!ok) return fail(UMGAP_ERR_IO, std::string(path) + ": truncated or unreadable table");
    return UMGAP_OK;
}

}  // namespace umgap
