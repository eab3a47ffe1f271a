// walk_ref.hpp — how a look-up reads the walk array of a walk-order 9-mer index (kind 10, index_walk.hip), shared by every kernel
// that does (gfx950 library internal).
//
// The table of a walk index maps a key to the number p of its entry E[p].  Table values are 32 bits, the real 9-mer index has 8 G
// keys and more: from 2^32 - 2^22 entries on (round 4) the table holds  q = p >> shift  (shift 1: 8.6 G entries, shift 2: 17 G) and
// the key's entry is the one of the 2^shift entries  E[q << shift ..]  that carries the key's low 31 bits.  That is exact, not a
// fingerprint test: the table said the key is present, so one of those entries IS its entry, and the builder lists the keys for
// which a neighbour in the group shows the same low bits under another key -- homopolymer stretches, and chance at 2^-31 per pair
// -- in a sorted exception list (key -> p) that such a look-up consults.  The builder starts every fragment's run of entries at a
// multiple of 2^shift (padding entries: low key bits all ones, which no 9-mer over 'A'..'Z' has), so the entries of a group come
// from one fragment and the thread that writes them sees every such clash.
#pragma once
#include <cstdint>

#include "common.hpp"

namespace umgap {

struct WalkRef {
    const uint64_t* E = nullptr;          // taxon | (low 31 key bits | continues << 31) << 32
    uint64_t n = 0;
    uint32_t shift = 0;
    uint32_t n_exc = 0;
    uint32_t zero_absent = 0;             // an entry with taxon 0 is a 9-mer of the proteome that is no key: answered with a miss
    const uint64_t* exc_keys = nullptr;   // sorted packed keys whose group holds another entry with the same low bits
    const uint64_t* exc_pos = nullptr;    // their positions
};
constexpr uint64_t WALK_NOPOS = ~0ull;
constexpr uint64_t WALK_PAD_ENTRY = 0x7FFFFFFF00000000ull;   // no taxon, low key bits all ones, continues nothing

inline WalkRef walk_ref_of(const umgap_index* ix) {
    WalkRef W;
    W.E = ix->d_walk;
    W.n = ix->n_walk;
    W.shift = ix->walk_shift;
    W.n_exc = ix->n_walk_exc;
    W.zero_absent = ix->walk_zero_absent ? 1u : 0u;
    W.exc_keys = ix->d_walk_exc_keys;
    W.exc_pos = ix->d_walk_exc_pos;
    return W;
}

// position of `key`'s entry given the table's answer q (UMGAP_NONE: the key is not in the index)
__device__ __forceinline__ uint64_t walk_position(const WalkRef& W, uint32_t q, uint64_t key) {
    if (q == UMGAP_NONE) return WALK_NOPOS;
    if (W.shift == 0) return q < W.n ? (uint64_t)q : WALK_NOPOS;
    const uint64_t base = (uint64_t)q << W.shift;
    const uint32_t want = (uint32_t)key & 0x7FFFFFFFu, g = 1u << W.shift;
    uint64_t found = WALK_NOPOS;
    uint32_t hits = 0;
    for (uint32_t i = 0; i < g; i++) {
        const uint64_t p = base + i;
        if (p >= W.n) break;
        if (((uint32_t)(W.E[p] >> 32) & 0x7FFFFFFFu) == want) { if (!hits) found = p; hits++; }
    }
    if (hits > 1 && W.n_exc) {          // another key of the group shows the same low bits: the list knows which entry is this key's
        uint32_t lo = 0, hi = W.n_exc;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            const uint64_t k = W.exc_keys[mid];
            if (k == key) { const uint64_t p = W.exc_pos[mid]; return p < W.n ? p : WALK_NOPOS; }   // (the loader checks the list; a second lock)
            if (k < key) lo = mid + 1; else hi = mid;
        }
    }
    return found;                        // (several entries of the SAME key -- a kept repeat behind the listed window: the first is the listed one)
}
// the taxon of that entry, or `miss`
__device__ __forceinline__ uint32_t walk_taxon(const WalkRef& W, uint32_t q, uint64_t key, uint32_t miss) {
    const uint64_t p = walk_position(W, q, key);
    if (p == WALK_NOPOS) return miss;
    const uint32_t t = (uint32_t)W.E[p];
    return (t == 0u && W.zero_absent) ? miss : t;
}

}  // namespace umgap
