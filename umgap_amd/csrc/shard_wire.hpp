// shard_wire.hpp — the wire format of the sharded lookup (shard.hip) and the device helpers that read it
// (gfx950 library internal).
//
// Out (keys): a key travels as h = mix45(key45), 45 bits in 6 bytes.  The owner derives the home bucket and the stored
// remainder straight from h (common.hpp), so nothing is hashed twice.  A send buffer holds `world` regions of `cap`
// slots; region o = [cap x u32: h & 0xFFFFFFFF][cap x u16: h >> 32] = 6*cap contiguous bytes = one message to peer o.
// A u16 of 0xFFFF marks a slot that holds no key (the unused tail of a region).
// Back (values): one value per slot, same slot order, `value_bytes` = 3 or 4 bytes each (3 when every value of every
// shard is below 2^24 - 1, which holds for NCBI taxon ids); all-ones = the key is absent.  3-byte values are packed four
// to three u32 words (cap is a multiple of 4).
//
// Where a key's slot is (round 3: no per-key permutation in memory).  A piece's keys are cut into RUNS of `run` consecutive
// keys -- the reads one wave of the consuming kernel takes together (a group of the fused classify kernels; 1024 keys for the
// plain get()).  A wave of the pack takes the same run, walks it in rounds of 64 keys (round r, lane l = key 64 r + l) and gives the
// keys of owner o the consecutive slots  base[run][o] + (number of earlier keys of the run with that owner)  of region o;
// base[run][o] comes from one atomic per workgroup and owner and is the only thing stored: `world` words per run (32 bytes per
// 1008 keys at 8 ranks) instead of 4 bytes per key.  The consumer walks the run in the same order, recomputes owner and rank of
// every key (run_rank below: the same function on both sides) and reads its value at base + rank: the values of a run lie in
// `world` short contiguous stretches (~126 values each at 8 ranks), so the gather is a handful of coalesced requests per
// round instead of one random sector per key.  A rank's OWN share (1/N of the keys) is never packed: the consumer has the key
// and its hash in hand and probes the rank's shard itself.
#pragma once
#include <cstdint>

#include "common.hpp"

namespace umgap {

struct HitSource {
    const uint32_t* ret = nullptr;    // returned values, slot order (region o at o * cap)
    const uint32_t* bases = nullptr;  // [run][world]: first slot (within region o) of the run's keys owned by o
    const uint64_t* keys = nullptr;   // the piece's packed 9-mers
    uint32_t value_bytes = 4;
    uint32_t world = 1, self = 0;     // self = this rank, or wire::NO_SELF when the rank's own keys travel like the others
                                      // (UMGAP_SHARD_NO_LOCAL=1: region `self` is then packed, probed and gathered too)
    uint32_t run = 1024;              // keys per run
    uint64_t cap = 0;                 // slots per region
    // SRC_WALK (classify.hip): not an exchange at all -- the index is in walk order and these name its walk array
    const uint64_t* walk_E = nullptr;
    uint64_t walk_n = 0;
    uint32_t walk_shift = 0, walk_n_exc = 0;
    const uint64_t* walk_exc_keys = nullptr;
    const uint64_t* walk_exc_pos = nullptr;
    uint32_t walk_debug = 0;              // timing experiments only (UMGAP_WALK_DEBUG; answers are WRONG when set): 1 no phase B, 2 no entry loads, 4 no anchor probe
};
// where the hits of a group of the fused 9-mer kernels come from
constexpr int SRC_PROBE = 0, SRC_GATHER = 1, SRC_WALK = 2;

// reads a wave of the fused classify kernels takes together (classify.hip): the run of the exchange is that many reads
inline uint32_t classify9_group_reads(uint32_t kmers_per_read, uint32_t fragments_per_read) {
    uint32_t G = 1024u / kmers_per_read;
    if (fragments_per_read && G * fragments_per_read > 64) G = 64 / fragments_per_read;
    return G;
}

namespace wire {

constexpr uint32_t NO_OWNER = 0xFFFFFFFFu;   // not a packed 9-mer: never sent, answered with a miss
constexpr uint32_t NO_SELF = 0xFFFFFFFEu;    // "this rank keeps no key to itself" (UMGAP_SHARD_NO_LOCAL=1): equals no owner, NO_OWNER included
constexpr uint32_t HI_NOKEY = 0xFFFFu;

__host__ __device__ inline uint32_t miss_raw(uint32_t value_bytes) { return value_bytes == 3 ? 0xFFFFFFu : 0xFFFFFFFFu; }

// shard s owns home buckets [nb*s/world, nb*(s+1)/world)   (alloc_table in index.hip).  Exact without a division: an
// estimate from the double product, put right by the two boundary conditions  nb*s < (b+1)*world <= nb*(s+1).
__host__ __device__ inline uint32_t owner_of_bucket(uint32_t b, uint32_t nb, uint32_t world, double world_over_nb) {
    uint32_t s = (uint32_t)((double)b * world_over_nb);
    if (s >= world) s = world - 1;
    const uint64_t t = ((uint64_t)b + 1) * world;
    while ((uint64_t)nb * s >= t) s--;
    while ((uint64_t)nb * (s + 1) < t) s++;
    return s;
}

// a key as both sides of the exchange see it
struct Routed {
    uint64_t h;        // mix45(key)
    uint32_t bucket;   // home bucket in the whole table
    uint32_t tag;      // stored remainder
    uint32_t owner;    // shard that holds the bucket, or NO_OWNER
};
__device__ __forceinline__ Routed route(uint64_t key, bool valid, uint32_t m, uint32_t nb, uint32_t world, double world_over_nb) {
    Routed r;
    r.h = mix45(key);
    r.tag = (uint32_t)r.h & R_MASK;
    r.bucket = (uint32_t)(r.h >> R_BITS) * m + (uint32_t)(((uint64_t)r.tag * m) >> R_BITS);
    r.owner = (valid && (key >> 45) == 0) ? owner_of_bucket(r.bucket, nb, world, world_over_nb) : NO_OWNER;
    return r;
}

// One round of 64 keys of a run: the rank of this lane's key among the run's keys with the same owner, in (round, lane)
// order.  `seen` lives in lane o: the number of keys of owner o in the earlier rounds; updated here.  Every lane of the
// wave calls this (ballots and a shuffle); a lane without a key passes NO_OWNER.  world <= 64.
__device__ __forceinline__ uint32_t run_rank(uint32_t owner, uint32_t world, uint32_t lane, uint32_t& seen) {
    const uint32_t prior = (uint32_t)__shfl((int)seen, (int)(owner & 63u), 64);
    const uint64_t below = (1ull << lane) - 1ull;
    uint32_t rank = 0;
    for (uint32_t o = 0; o < world; o++) {
        const uint64_t mask = __ballot(owner == o);
        if (owner == o) rank = prior + (uint32_t)__builtin_popcountll(mask & below);
        if (lane == o) seen += (uint32_t)__builtin_popcountll(mask);
    }
    return rank;
}

// raw value of a slot (miss_raw = absent)
__device__ __forceinline__ uint32_t load_raw(const uint32_t* __restrict__ ret, uint64_t slot, uint32_t value_bytes) {
    if (value_bytes == 4) return ret[slot];
    const uint32_t i = (uint32_t)slot & 3u;
    const uint32_t* w = ret + (size_t)(slot >> 2) * 3 + (i >> 1);
    const uint32_t lo = w[0], hi = w[1];
    const uint32_t v = i == 0 ? lo : (i == 1 ? ((lo >> 24) | (hi << 8)) : (i == 2 ? ((lo >> 16) | (hi << 16)) : (hi >> 8)));
    return v & 0xFFFFFFu;
}
// taxon of a slot as the aggregators want it: 0 = miss (what `prot2kmer2lca -o` prints, src/commands/prot2kmer2lca.rs:176-178)
__device__ __forceinline__ uint32_t load_value(const uint32_t* __restrict__ ret, uint64_t slot, uint32_t value_bytes) {
    const uint32_t v = load_raw(ret, slot, value_bytes);
    return v == miss_raw(value_bytes) ? 0u : v;
}

}  // namespace wire
}  // namespace umgap
