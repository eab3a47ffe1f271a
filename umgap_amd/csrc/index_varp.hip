// index_varp.hip — the PACKED format of the variable-length (tryptic peptide) index: build, probe, dump (gfx950 only).
// Layout and rationale: var_index.hpp.  What it replaces: fst::Map::get on the tryptic index (src/commands/pept2lca.rs:93,
// src/commands/prot2tryp2lca.rs:130; keys of 5..50 letters over 'A'..'Z', scripts/umgap-analyse.sh:291).
//
// The legacy format (index_var.hip) keeps 16-byte slots {tag, taxon, heap offset} and verifies a candidate against the key bytes
// in a heap: a hit costs the bucket plus one or two heap sectors, and at the chip's random-request ceiling (~50 G sectors/s,
// DESIGN.md section 4) that is all that matters: 1.5 requests per lookup on insertion-ordered queries, 2.1 on shuffled ones.  Here the
// key sits in the bucket itself (1-3 slots at 5 bits per residue), so every lookup, hit or miss, is ONE request unless a key homed
// in the same bucket was displaced at build time.
//
//   k_varp_scan      are all keys words over 'A'..'Z' of 1..63 letters?  how many slots do they take?         (build)
//   k_varp_insert    key -> images -> home bucket; S adjacent slots reserved by a compare-and-swap on the bucket's fill count
//   k_lookup_varp    the probe: a batch of 64 queries per wave and step; bytes staged in LDS by the pipeline of the legacy kernel
//                    (offsets two batches ahead, bytes one batch ahead through global_load_lds), every lane packs its own query
//                    from LDS four residues per word, a quad reads a bucket, whole-slot compares, DPP inside the quad
#include <algorithm>
#include <cmath>
#include <type_traits>
#include <vector>

#include "var_index.hpp"

using namespace umgap;
using namespace umgap::probe;

namespace {

struct VarpFlags {
    unsigned long long bad_len, not_az, slots, n_stash, too_full, mismatch;
};
// stash capacity of a build: a few keys in ten thousand find no room within VARP_MAX_DISP buckets at the default fill (0.8 per
// thousand of 1 G keys of 14-45 letters), so it grows with the table
inline size_t varp_stash_cap(size_t n) { return std::max<size_t>((size_t)1 << 20, n / 256); }

__global__ __launch_bounds__(256) void k_varp_scan(const uint8_t* __restrict__ heap, const uint64_t* __restrict__ off, size_t n, VarpFlags* flags) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    unsigned long long slots = 0, bad_len = 0, not_az = 0;
    for (; i < n; i += step) {
        const uint64_t o = off[i], l64 = off[i + 1] - o;
        if (l64 < 1 || l64 > VAR_MAX_LEN) { bad_len++; continue; }
        const uint32_t len = (uint32_t)l64;
        bool az = true;
        for (uint32_t k = 0; k < len; k++) az = az && (uint32_t)(heap[o + k] - 'A') < 26u;
        if (!az) not_az++;
        slots += varp_slots(len);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        slots += ((unsigned long long)(uint32_t)__shfl_xor((int)(uint32_t)(slots >> 32), d, 64) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)slots, d, 64);
        bad_len += (uint32_t)__shfl_xor((int)(uint32_t)bad_len, d, 64);      // per-lane counts stay far below 2^32
        not_az += (uint32_t)__shfl_xor((int)(uint32_t)not_az, d, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        if (slots) atomicAdd(&flags->slots, slots);
        if (bad_len) atomicAdd(&flags->bad_len, bad_len);
        if (not_az) atomicAdd(&flags->not_az, not_az);
    }
}

// `used`: 16-bit fill count per bucket (two per word), build time only.  An entry never crosses a bucket: its S slots are
// reserved by a compare-and-swap on the count (first fit in arrival order: a bucket that cannot take a three-slot entry still
// takes a later one-slot entry).  Where a displaced key went is recorded in slot 0 of its HOME bucket (var_index.hpp).
__global__ __launch_bounds__(256) void k_varp_insert(u32x4* __restrict__ slots, uint32_t* __restrict__ used, uint64_t nb, const uint8_t* __restrict__ heap,
                                                     const uint64_t* __restrict__ off, const uint32_t* __restrict__ vals, size_t n, VarpFlags* flags,
                                                     VarpStashEntry* __restrict__ stash, unsigned long long stash_cap) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += step) {
        const uint64_t o = off[i], l64 = off[i + 1] - o;
        if (l64 < 1 || l64 > VAR_MAX_LEN) { atomicAdd(&flags->bad_len, 1ull); continue; }
        VarpKey K;
        if (!varp_pack_bytes(heap + o, (uint32_t)l64, K)) { atomicAdd(&flags->not_az, 1ull); continue; }
        const uint32_t S = varp_slots((uint32_t)l64);
        const uint32_t h = varp_hash(K);
        const uint64_t home = varp_home(h, nb);
        uint32_t* home_w = reinterpret_cast<uint32_t*>(slots + home * VAR_SLOTS) + 3;
        bool placed = false;
        for (uint32_t d = 0; d <= VARP_MAX_DISP && !placed; d++) {
            const uint64_t b = home + d;
            uint32_t* cnt = used + (b >> 1);
            const uint32_t sh = (uint32_t)(b & 1) * 16;
            uint32_t cur = *reinterpret_cast<volatile uint32_t*>(cnt), old = 0;
            for (;;) {
                old = (cur >> sh) & 0xFFFFu;
                if (old + S > VAR_SLOTS) break;
                const uint32_t seen = atomicCAS(cnt, cur, cur + (S << sh));
                if (seen == cur) { placed = true; break; }
                cur = seen;
            }
            if (placed) {
                uint32_t* s = reinterpret_cast<uint32_t*>(slots + b * VAR_SLOTS) + 4 * old;
                s[0] = vals[i];
                s[1] = K.q[0][1];
                s[2] = K.q[0][2];
                atomicOr(s + 3, K.q[0][3]);     // OR, not store: slot 0's w also collects the marks of the keys displaced from this bucket
                for (uint32_t c = 1; c < S; c++) {
                    s[4 * c] = K.q[c][0]; s[4 * c + 1] = K.q[c][1]; s[4 * c + 2] = K.q[c][2]; s[4 * c + 3] = K.q[c][3];
                }
                if (d) atomicOr(home_w, 1u << (VARP_DISP_SHIFT + d));
            }
        }
        if (!placed) {
            const unsigned long long at = atomicAdd(&flags->n_stash, 1ull);
            if (at < stash_cap) {
                VarpStashEntry E;
                E.hash = h;
                E.img[0] = K.q[0][1]; E.img[1] = K.q[0][2]; E.img[2] = K.q[0][3];
                for (int c = 0; c < 2; c++)
                    for (int t = 0; t < 4; t++) E.img[3 + 4 * c + t] = K.q[1 + c][t];
                E.taxon = vals[i];
                stash[at] = E;
                atomicOr(home_w, VARP_STASHED);
            } else atomicAdd(&flags->too_full, 1ull);
        }
    }
}

__global__ void k_varp_selfcheck(const uint32_t* __restrict__ got, const uint32_t* __restrict__ want, size_t n, VarpFlags* flags) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride)
        if (got[i] != want[i]) atomicAdd(&flags->mismatch, 1ull);
}

// ---- the probe ----------------------------------------------------------------------------------------------------------------
// four key bytes (one little-endian word, nv of them valid) -> their 20-bit group; a valid byte outside 0x40..0x5F raises `bad`.
// (Round 5: the test is the two instructions "is the byte in the 32 codes around the capitals", not the six of the exact 'A'..'Z' test
// of pack_inline_words.  The six bytes it lets through beside the capitals -- '@' and '[' .. '_' -- come out as residue codes 31 and
// 26 .. 30, which no key of the table holds (the builder takes 'A'..'Z' only, varp_pack_bytes): a query with such a byte cannot equal a
// key in the slot that holds it and is a miss like before, through a probe instead of without one.  A borrow of the subtraction below
// can only change the codes of a query that already has such a byte.)
__device__ __forceinline__ uint32_t varp_group(uint32_t x, int nv, uint32_t& bad) {
    const uint32_t mask = nv >= 4 ? 0xFFFFFFFFu : (nv <= 0 ? 0u : (0xFFFFFFFFu >> (32 - 8 * nv)));
    bad |= (x ^ 0x40404040u) & 0xE0E0E0E0u & mask;
    const uint32_t c = (x - 0x41414141u) & mask;
    const uint32_t t = (c & 0x001F001Fu) | ((c & 0x1F001F00u) >> 3);
    return (t & 0x3FFu) | ((t >> 6) & 0xFFC00u);
}
// the lane's query from the staged bytes (w = the wave's LDS strip, s = byte offset of the key in it); false: not a word over 'A'..'Z'
__device__ __forceinline__ bool varp_pack_lds(const uint32_t* w, uint32_t s, uint32_t len, VarpKey& K) {
    const uint32_t k = s >> 2, sh = s & 3u;
    uint32_t g[16], bad = 0;
    uint32_t prev = w[k];
#pragma unroll
    for (int i = 0; i < 16; i++) g[i] = 0;
#pragma unroll
    for (int c = 0; c < 4; c++) {       // four words at a time, as far as the longest key of the wave goes (wave-uniform)
        if (c && !__any(len > 16u * c)) break;
#pragma unroll
        for (int i = 4 * c; i < 4 * c + 4; i++) {
            const uint32_t nxt = w[k + i + 1];
            g[i] = varp_group(__builtin_amdgcn_alignbyte(nxt, prev, sh), (int)len - 4 * i, bad);
            prev = nxt;
        }
    }
    varp_images(g, bad ? 0u : len, K);
    return bad == 0;
}

// A quad holds a bucket (lane j: slot j = e).  Does it hold the query -- header image (y, z, w), continuation images c1, c2, S slots?
// Returns bit 0 = found (val = its taxon, quad-uniform); bits 23..30 = where the keys displaced from THIS bucket went (slot 0's marks).
// (whole-slot compares as xor / or chains: one v_cmp per image instead of four and no scalar and-chains)
__device__ __forceinline__ uint32_t varp_eval(u32x4 e, uint32_t qy, uint32_t qz, uint32_t qw, const uint32_t (&c1)[4], const uint32_t (&c2)[4], uint32_t S,
                                              uint32_t j, uint32_t& val) {
    uint32_t f = (((e.y ^ qy) | (e.z ^ qz) | ((e.w & ~VARP_WHERE) ^ qw)) == 0u) ? 1u : 0u;       // qw carries the header mark: a continuation slot never matches
    if (S >= 2u) {      // quad-uniform
        const uint32_t m1 = (((e.x ^ c1[0]) | (e.y ^ c1[1]) | (e.z ^ c1[2]) | (e.w ^ c1[3])) == 0u) ? 1u : 0u;
        const uint32_t n1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m1, 0xF9, 0xF, 0xF, true);   // quad_perm [1,2,3,3]: the next slot's m1
        f &= j < 3u ? n1 : 0u;
        if (S >= 3u) {
            const uint32_t m2 = (((e.x ^ c2[0]) | (e.y ^ c2[1]) | (e.z ^ c2[2]) | (e.w ^ c2[3])) == 0u) ? 1u : 0u;
            const uint32_t n2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m2, 0xFE, 0xF, 0xF, true);   // quad_perm [2,3,3,3]: the slot after's m2
            f &= j < 2u ? n2 : 0u;
        }
    }
    val = quad_or(f ? e.x : 0u);
    return quad_or(f | (j == 0u ? (e.w & VARP_WHERE) : 0u));
}

__device__ __noinline__ uint32_t varp_stash_lookup(const VarView T, uint32_t h, const VarpKey Q, uint32_t miss) {
    const uint32_t S = varp_slots(Q.q[0][1] & 63u);
    uint32_t lo = 0, hi = T.n_stash;
    while (lo < hi) {  // first entry with hash >= h
        const uint32_t mid = (lo + hi) >> 1;
        if (T.pstash[mid].hash < h) lo = mid + 1; else hi = mid;
    }
    for (; lo < T.n_stash && T.pstash[lo].hash == h; lo++) {
        const VarpStashEntry& E = T.pstash[lo];
        bool same = E.img[0] == Q.q[0][1] && E.img[1] == Q.q[0][2] && E.img[2] == Q.q[0][3];
        for (int t = 0; t < 4; t++) {
            same = same && (S < 2u || E.img[3 + t] == Q.q[1][t]);
            same = same && (S < 3u || E.img[7 + t] == Q.q[2][t]);
        }
        if (same) return E.taxon;
    }
    return miss;
}

// lane r of the caller's quad (r a compile-time constant): a DPP move, no LDS traffic, legal under quad-uniform branches
template <int R>
__device__ __forceinline__ uint32_t quad_bcast(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, R | (R << 2) | (R << 4) | (R << 6), 0xF, 0xF, true);   // (old = 0 + bound_ctrl: the form the compiler folds into the consuming instruction)
}

// One batch of 64 packed queries (K: the lane's own; K.q[0][1] == 0: no probe, a miss).  A quad serves the four queries of its
// own four lanes, one per round: round r = the query of lane 4q + r, its images and bucket number broadcast inside the quad by DPP.
// The four home buckets are fetched back to back.  A query that its home bucket does not hold but whose home bucket has displaced
// keys goes on to exactly the buckets the marks name: those loads are issued together for all four rounds, one displacement per
// step (one step is the rule: a bucket rarely displaces to two distances), then the stash if it is marked.
__device__ __forceinline__ void varp_probe_batch(const VarView& T, const VarpKey& K, size_t base, size_t n, uint32_t miss, uint32_t* __restrict__ out, uint32_t lane) {
    const uint32_t j = lane & 3;
    const uint32_t hash = varp_hash(K);
    const uint64_t bucket = K.q[0][1] ? varp_home(hash, T.nb) : 0ull;
    uint32_t where[4], res[4];
    u32x4 e[4];
    uint64_t rb[4];
    auto round_bucket = [&](auto R) {
        constexpr int r = decltype(R)::value;
        rb[r] = (uint64_t)quad_bcast<r>((uint32_t)bucket) | ((uint64_t)quad_bcast<r>((uint32_t)(bucket >> 32)) << 32);
        e[r] = quad_bcast<r>(K.q[0][1]) ? __builtin_nontemporal_load(T.slots + rb[r] * VAR_SLOTS + j) : u32x4{0u, 0u, 0u, 0u};
    };
    round_bucket(std::integral_constant<int, 0>{});
    round_bucket(std::integral_constant<int, 1>{});
    round_bucket(std::integral_constant<int, 2>{});
    round_bucket(std::integral_constant<int, 3>{});
    // evaluates bucket e[r] against round r's query; returns varp_eval's word (0 for "no query")
    auto eval_round = [&](auto R, uint32_t& val) -> uint32_t {
        constexpr int r = decltype(R)::value;
        const uint32_t qy = quad_bcast<r>(K.q[0][1]);
        if (!qy) return 0u;      // quad-uniform
        const uint32_t S = varp_slots(qy & 63u);
        uint32_t c1[4] = {0u, 0u, 0u, 0u}, c2[4] = {0u, 0u, 0u, 0u};
        if (S >= 2u) {
#pragma unroll
            for (int t = 0; t < 4; t++) c1[t] = quad_bcast<r>(K.q[1][t]);
        }
        if (S >= 3u) {
#pragma unroll
            for (int t = 0; t < 4; t++) c2[t] = quad_bcast<r>(K.q[2][t]);
        }
        return varp_eval(e[r], qy, quad_bcast<r>(K.q[0][2]), quad_bcast<r>(K.q[0][3]), c1, c2, S, j, val);
    };
    uint32_t more = 0;     // quad-uniform: some round of this quad has further buckets (or the stash) to look at
    auto first = [&](auto R) {
        constexpr int r = decltype(R)::value;
        uint32_t val = 0;
        const uint32_t f = eval_round(R, val);
        res[r] = (f & 1u) ? val : miss;
        where[r] = (f & 1u) ? 0u : (f & VARP_WHERE);
        more |= where[r];
    };
    first(std::integral_constant<int, 0>{});
    first(std::integral_constant<int, 1>{});
    first(std::integral_constant<int, 2>{});
    first(std::integral_constant<int, 3>{});
    while (__any(more != 0u)) {     // wave-uniform; rare beyond one trip
        uint32_t d[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const uint32_t marks = (where[r] & ~VARP_STASHED) >> (VARP_DISP_SHIFT + 1);      // bit d - 1 = displacement d
            d[r] = marks ? (uint32_t)__builtin_ctz(marks) + 1u : 0u;
            if (d[r]) e[r] = __builtin_nontemporal_load(T.slots + (rb[r] + d[r]) * VAR_SLOTS + j);
        }
        more = 0;
        auto next = [&](auto R) {
            constexpr int r = decltype(R)::value;
            if (!where[r]) return;          // quad-uniform
            if (d[r]) {
                uint32_t val = 0;
                const uint32_t f = eval_round(R, val);
                if (f & 1u) { res[r] = val; where[r] = 0u; }
                else where[r] &= ~(1u << (VARP_DISP_SHIFT + d[r]));
            } else {                        // every marked bucket looked at and the stash is marked
                VarpKey Q;
#pragma unroll
                for (int c = 0; c < 3; c++)
#pragma unroll
                    for (int t = 0; t < 4; t++) Q.q[c][t] = quad_bcast<r>(K.q[c][t]);
                res[r] = varp_stash_lookup(T, quad_bcast<r>(hash), Q, miss);
                where[r] = 0u;
            }
            more |= where[r];
        };
        next(std::integral_constant<int, 0>{});
        next(std::integral_constant<int, 1>{});
        next(std::integral_constant<int, 2>{});
        next(std::integral_constant<int, 3>{});
    }
    // lane 4q + r owns the query of round r
    const size_t k = base + lane;
    if (k < n) out[k] = j == 0 ? res[0] : (j == 1 ? res[1] : (j == 2 ? res[2] : res[3]));
}

// a batch that cannot be staged (ill-formed offsets): every lane packs its key from global memory
__device__ __noinline__ VarpKey varp_pack_global(const uint8_t* __restrict__ qbytes, const uint64_t* __restrict__ qoff, size_t n, size_t base, uint32_t lane,
                                                 uint32_t min_len, uint32_t max_len) {
    VarpKey K;
    for (int c = 0; c < 3; c++)
        for (int t = 0; t < 4; t++) K.q[c][t] = 0u;
    const size_t i = base + lane;
    if (i < n) {
        const uint64_t o = qoff[i], l64 = qoff[i + 1] - o;
        if (qoff[i + 1] >= o && l64 >= min_len && l64 <= max_len) {
            if (!varp_pack_bytes(qbytes + o, (uint32_t)l64, K)) K.q[0][1] = 0u;
        }
    }
    return K;
}

// (Round 5: a fifth wave per SIMD -- strips cut to 4096 bytes, 32 KB per workgroup, 96 registers -- measured 6 % SLOWER on the same box
// (the tryptic call 3.05-3.08 against 2.85-2.91 ms per 10 M pairs): like the 9-mer probes, this one wants four.)
// The kernel: the pipeline of k_lookup_var_pipe (index_var.hip) -- while batch t probes, the bytes of batch t + 1 travel from global
// memory straight into the other LDS strip and the offsets of batch t + 2 are in flight -- with the packed probe in it.
__global__ __launch_bounds__(256, 4) void k_lookup_varp(const uint8_t* __restrict__ qbytes, const uint64_t* __restrict__ qoff, size_t n, uint32_t miss,
                                                        uint32_t* __restrict__ out, VarView T, uint32_t min_len, uint32_t max_len) {
    // (+ 16 words per strip: the packer reads four words at a time as far as the LONGEST key of the wave goes, from every lane's key start)
    __shared__ __attribute__((aligned(16))) uint32_t lds[4][2][VAR_STAGE / 4 + 16];
    const uint32_t wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t step = (((size_t)gridDim.x * blockDim.x) >> 6) * 64;
    size_t base = wave * 64;
    if (base >= n) return;
    const uint64_t all_end = qoff[n];
    const uint64_t lo_abs = (uint64_t)(uintptr_t)qbytes, hi_abs = (uint64_t)(uintptr_t)(qbytes + all_end);
    auto load_offsets = [&](size_t b, uint64_t& o, uint64_t& oe) {
        o = qoff[min(b + lane, n)];
        oe = qoff[min(b + 64, n)];
    };
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
        __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): the strip of this batch and the offsets of the next were issued an iteration ago
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const uint32_t* w = lds[wv][t & 1u];
        VarpKey K;
        if (S_cur.ok) {
            const uint32_t s = meta_cur & 0x1FFFu, len = (meta_cur >> 13) & 63u;
            bool ok = (meta_cur >> 19) & 1u;
            if (ok) ok = varp_pack_lds(w, s, len, K);
            if (!ok) {
#pragma unroll
                for (int c = 0; c < 3; c++)
#pragma unroll
                    for (int x = 0; x < 4; x++) K.q[c][x] = 0u;
            }
        } else K = varp_pack_global(qbytes, qoff, n, base, lane, min_len, max_len);
        VarStage S_next{0, 0, false};
        uint32_t meta_next = 0;
        if (next < n) S_next = start_stage(o_next, oe_next, (t + 1u) & 1u, meta_next);
        load_offsets(next2, o_next, oe_next);
        varp_probe_batch(T, K, base, n, miss, out, lane);
        if (next >= n) break;
        base = next;
        S_cur = S_next;
        meta_cur = meta_next;
    }
}

inline int varp_grid(size_t n, size_t per_cu) {
    size_t g = (n + 255) / 256;
    if (g < 1) g = 1;
    if (g > 256 * per_cu) g = 256 * per_cu;
    return (int)g;
}

}  // namespace

namespace umgap {

// Are these keys fit for the packed format, and how many slots would they take?  (synchronises `stream`)
int varp_scan(const uint8_t* d_bytes, const uint64_t* d_off, size_t n, hipStream_t stream, uint64_t* bad_len, uint64_t* not_az, uint64_t* slots) {
    VarpFlags* d_flags = nullptr;
    UMGAP_HIP(hipMalloc((void**)&d_flags, sizeof(VarpFlags)));
    hipError_t e = hipMemsetAsync(d_flags, 0, sizeof(VarpFlags), stream);
    if (e == hipSuccess && n) hipLaunchKernelGGL(k_varp_scan, dim3(varp_grid(n, 8)), dim3(256), 0, stream, d_bytes, d_off, n, d_flags);
    VarpFlags h{};
    if (e == hipSuccess) e = hipMemcpyAsync(&h, d_flags, sizeof(h), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    (void)hipFree(d_flags);
    if (e != hipSuccess) return fail(UMGAP_ERR_DEVICE, std::string("var index scan: ") + hipGetErrorString(e));
    *bad_len = h.bad_len; *not_az = h.not_az; *slots = h.slots;
    return UMGAP_OK;
}

int varp_lookup_dev(const umgap_var_index* v, const uint8_t* d_bytes, const uint64_t* d_off, size_t n, uint32_t miss, uint32_t* d_out, hipStream_t s,
                    uint32_t min_len, uint32_t max_len) {
    static int per_cu = 0;
    if (!per_cu) {
        int b = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, k_lookup_varp, 256, 0) != hipSuccess || b < 1) b = 4;
        per_cu = 2 * b;          // two resident rounds, as the legacy kernel
    }
    if (n) hipLaunchKernelGGL(k_lookup_varp, dim3(varp_grid(n, (size_t)per_cu)), dim3(256), 0, s, d_bytes, d_off, n, miss, d_out, v->view(), min_len, max_len);
    UMGAP_HIP(hipGetLastError());
    return UMGAP_OK;
}

// Builds the packed table of `v` (slots, stash) from the caller's keys; `total_slots` from varp_scan.  The key bytes are only read.
int varp_build(umgap_var_index* v, const uint8_t* d_bytes, const uint64_t* d_off, const uint32_t* d_vals, size_t n, uint64_t total_slots, double load,
               hipStream_t stream) {
    // default fill 0.4 of the slots: entries of 1-3 slots in buckets of 4 leave holes, and past ~0.45 more keys than the stash takes
    // find no room within VARP_MAX_DISP buckets (1 G keys of 14-45 letters: 82.5 GB of table, 7 % of the keys displaced)
    if (!(load > 0.0 && load < 1.0)) load = 0.4;
    v->inline_max = VARP_INLINE_ALL;
    v->n_keys = n;
    VarpFlags* d_flags = nullptr;
    VarpStashEntry* d_stash = nullptr;
    uint32_t *d_got = nullptr, *d_used = nullptr;
    int rc = UMGAP_OK;
    auto cleanup = [&]() { for (void* p : {(void*)d_flags, (void*)d_stash, (void*)d_got, (void*)d_used}) if (p) (void)hipFree(p); };
    if (hipMalloc((void**)&d_flags, sizeof(VarpFlags)) != hipSuccess || hipMalloc((void**)&d_stash, varp_stash_cap(n) * sizeof(VarpStashEntry)) != hipSuccess ||
        hipMalloc((void**)&d_got, std::max<size_t>(n, 1) * 4) != hipSuccess) { cleanup(); return fail(UMGAP_ERR_NOMEM, "hipMalloc(var index build)"); }
    uint64_t nb = std::max<uint64_t>(1024, (uint64_t)std::ceil((double)total_slots / (VAR_SLOTS * load)));
    for (int attempt = 0; attempt < 4; attempt++) {
        if (nb >= (1ull << 32)) { rc = fail(UMGAP_ERR_BUILD, "index build: the packed layout holds fewer than 2^32 buckets (256 GB of table)"); break; }
        if (v->d_slots) { (void)hipFree(v->d_slots); v->d_slots = nullptr; }
        if (d_used) { (void)hipFree(d_used); d_used = nullptr; }
        const size_t nbuckets = (size_t)(nb + VAR_MAX_DISP), used_words = nbuckets / 2 + 2;
        if (hipMalloc((void**)&v->d_slots, nbuckets * 64) != hipSuccess || hipMalloc((void**)&d_used, used_words * 4) != hipSuccess) { rc = fail(UMGAP_ERR_NOMEM, "hipMalloc of the var table failed"); break; }
        v->nb = nb;
        hipError_t e = hipMemsetAsync(v->d_slots, 0, nbuckets * 64, stream);
        if (e == hipSuccess) e = hipMemsetAsync(d_used, 0, used_words * 4, stream);
        if (e == hipSuccess) e = hipMemsetAsync(d_flags, 0, sizeof(VarpFlags), stream);
        if (e == hipSuccess && n) hipLaunchKernelGGL(k_varp_insert, dim3(varp_grid(n, 8)), dim3(256), 0, stream, v->d_slots, d_used, nb, d_bytes, d_off, d_vals, n, d_flags, d_stash, (unsigned long long)varp_stash_cap(n));
        VarpFlags h{};
        if (e == hipSuccess) e = hipMemcpyAsync(&h, d_flags, sizeof(h), hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        if (e != hipSuccess) { rc = fail(UMGAP_ERR_DEVICE, std::string("var index build: ") + hipGetErrorString(e)); break; }
        if (h.bad_len) { rc = fail(UMGAP_ERR_BUILD, "index build: " + std::to_string(h.bad_len) + " keys are empty or longer than 63 bytes"); break; }
        if (h.not_az) { rc = fail(UMGAP_ERR_BUILD, "index build: the packed format takes keys over 'A'..'Z' only"); break; }
        if (h.too_full) { nb = nb + nb / 4 + 1; rc = fail(UMGAP_ERR_BUILD, "index build: table too full"); continue; }
        v->n_stash = (uint32_t)h.n_stash;
        if (v->d_pstash) { (void)hipFree(v->d_pstash); v->d_pstash = nullptr; }
        if (h.n_stash) {
            std::vector<VarpStashEntry> st(h.n_stash);
            if (hipMemcpy(st.data(), d_stash, st.size() * sizeof(VarpStashEntry), hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(UMGAP_ERR_DEVICE, "stash copy failed"); break; }
            std::sort(st.begin(), st.end(), [](const VarpStashEntry& a, const VarpStashEntry& b) { return a.hash < b.hash; });
            if (hipMalloc((void**)&v->d_pstash, st.size() * sizeof(VarpStashEntry)) != hipSuccess ||
                hipMemcpy(v->d_pstash, st.data(), st.size() * sizeof(VarpStashEntry), hipMemcpyHostToDevice) != hipSuccess) { rc = fail(UMGAP_ERR_NOMEM, "hipMalloc(var stash)"); break; }
        }
        // self check: every key finds its own taxon (a duplicate key finds the other copy's)
        if (n) {
            rc = varp_lookup_dev(v, d_bytes, d_off, n, 0xFFFFFFFFu, d_got, stream, 1u, VAR_MAX_LEN);
            if (rc) break;
            hipLaunchKernelGGL(k_varp_selfcheck, dim3(varp_grid(n, 8)), dim3(256), 0, stream, d_got, d_vals, n, d_flags);
            e = hipMemcpyAsync(&h, d_flags, sizeof(h), hipMemcpyDeviceToHost, stream);
            if (e == hipSuccess) e = hipStreamSynchronize(stream);
            if (e != hipSuccess) { rc = fail(UMGAP_ERR_DEVICE, std::string("var index self-check: ") + hipGetErrorString(e)); break; }
            if (h.mismatch) { rc = fail(UMGAP_ERR_BUILD, "index build: duplicate keys (keys must be distinct, cf. fst::MapBuilder::insert)"); break; }
        }
        if (getenv("UMGAP_TIMING"))
            fprintf(stderr, "packed var table: %zu keys in %llu slots, %llu buckets (fill %.3f), %llu keys in the stash (attempt %d)\n", n,
                    (unsigned long long)total_slots, (unsigned long long)nb, (double)total_slots / (4.0 * (double)nb), (unsigned long long)h.n_stash, attempt + 1);
        rc = UMGAP_OK;
        break;
    }
    cleanup();
    return rc;
}

// printindex: every key of the packed table, its bytes appended to heap_out (when given and large enough)
int varp_dump(const umgap_var_index* v, uint8_t* heap_out, uint64_t heap_cap, uint64_t* offsets_out, uint32_t* lens_out, uint32_t* taxa_out, size_t cap,
              size_t* n_out, uint64_t* heap_bytes_out) {
    size_t n = 0;
    uint64_t appended = 0;
    auto emit = [&](const uint32_t* hdr_yzw_minus1, uint32_t taxon, const uint32_t* c1, const uint32_t* c2) {
        uint8_t key[64];
        const uint32_t len = varp_unpack(hdr_yzw_minus1, c1, c2, key);
        if (heap_out && heap_cap >= appended + len) memcpy(heap_out + appended, key, len);
        if (n < cap) { offsets_out[n] = appended; lens_out[n] = len; taxa_out[n] = taxon; }
        appended += len;
        n++;
    };
    const size_t total = (size_t)(v->nb + VAR_MAX_DISP), chunk = 1u << 20;   // buckets
    std::vector<uint32_t> buf(chunk * 16);
    for (size_t b0 = 0; b0 < total; b0 += chunk) {
        const size_t cnt = std::min(chunk, total - b0);
        UMGAP_HIP(hipMemcpy(buf.data(), v->d_slots + b0 * VAR_SLOTS, cnt * 64, hipMemcpyDeviceToHost));
        for (size_t b = 0; b < cnt; b++) {
            const uint32_t* s = buf.data() + 16 * b;
            for (uint32_t k = 0; k < VAR_SLOTS && (s[4 * k + 3] & VARP_MARK);) {
                const uint32_t S = varp_slots(s[4 * k + 1] & 63u);
                if (k + S > VAR_SLOTS) return fail(UMGAP_ERR_FORMAT, "packed var table: an entry crosses its bucket");
                emit(s + 4 * k, s[4 * k], S >= 2 ? s + 4 * (k + 1) : nullptr, S >= 3 ? s + 4 * (k + 2) : nullptr);   // (the marks in slot 0's w lie outside the key bits)
                k += S;
            }
        }
    }
    if (v->n_stash) {
        std::vector<VarpStashEntry> st(v->n_stash);
        UMGAP_HIP(hipMemcpy(st.data(), v->d_pstash, st.size() * sizeof(VarpStashEntry), hipMemcpyDeviceToHost));
        for (const VarpStashEntry& E : st) {
            const uint32_t hdr[4] = {0u, E.img[0], E.img[1], E.img[2]};
            const uint32_t S = varp_slots(E.img[0] & 63u);
            emit(hdr, E.taxon, S >= 2 ? E.img + 3 : nullptr, S >= 3 ? E.img + 7 : nullptr);
        }
    }
    *n_out = n;
    *heap_bytes_out = appended;
    return UMGAP_OK;
}

}  // namespace umgap
