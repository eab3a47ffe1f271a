// shard_wire.hpp — the wire format of the sharded lookup (shard.hip) and the device helpers that read it
// (gfx950 library internal).
//
// Out (keys): a key travels as h = mix45(key45), 45 bits in 6 bytes.  The owner derives the home bucket and the stored
// remainder straight from h (common.hpp), so nothing is hashed twice.  A send buffer holds `world` regions of `cap`
// slots; region o = [cap x u32: h & 0xFFFFFFFF][cap x u16: h >> 32] = 6*cap contiguous bytes = one message to peer o.
// A u16 of 0xFFFF marks a slot that holds no key (the unused tail of a region).
// A rank's OWN share (1/N of the keys) takes none of this: the pack only marks those keys -- one bit per key (local_bits), no
// perm entry -- and the kernel that consumes the hits probes them itself, between its gathers: no 6-byte copy, no value store,
// no gather and no 4-byte perm round trip for them.
// Back (values): one value per slot, same slot order, `value_bytes` = 3 or 4 bytes each (3 when every value of every
// shard is below 2^24 - 1, which holds for NCBI taxon ids); all-ones = the key is absent.  3-byte values are packed four
// to three u32 words (cap is a multiple of 4).
#pragma once
#include <cstdint>

namespace umgap {

struct HitSource {
    const uint32_t* ret = nullptr;   // returned values, slot order
    const uint32_t* perm = nullptr;  // key i of the batch -> slot; wire::NOSLOT = not sent (answered with a miss);
                                     // wire::LOCAL = the rank owns the key: probed where it is consumed, never packed
    uint32_t value_bytes = 4;
    const uint64_t* keys = nullptr;  // the batch's packed 9-mers (same indexing as perm): read again for the LOCAL keys
    const uint64_t* local_bits = nullptr;   // bit i set = key i is LOCAL: its perm entry was never written and is never read
};

namespace wire {

// is key i of the batch the rank's own?  (no bitmap: the perm entry says so)
__device__ __forceinline__ bool is_local_key(const HitSource& H, size_t i) { return H.local_bits && ((H.local_bits[i >> 6] >> (i & 63)) & 1ull); }

constexpr uint32_t NOSLOT = 0xFFFFFFFFu;
constexpr uint32_t LOCAL = 0xFFFFFFFEu;
constexpr uint32_t HI_NOKEY = 0xFFFFu;

__host__ __device__ inline uint32_t miss_raw(uint32_t value_bytes) { return value_bytes == 3 ? 0xFFFFFFu : 0xFFFFFFFFu; }

// raw value of a slot (miss_raw = absent)
__device__ __forceinline__ uint32_t load_raw(const uint32_t* __restrict__ ret, uint32_t slot, uint32_t value_bytes) {
    if (value_bytes == 4) return ret[slot];
    const uint32_t i = slot & 3u;
    const uint32_t* w = ret + (size_t)(slot >> 2) * 3 + (i >> 1);
    const uint32_t lo = w[0], hi = w[1];
    const uint32_t v = i == 0 ? lo : (i == 1 ? ((lo >> 24) | (hi << 8)) : (i == 2 ? ((lo >> 16) | (hi << 16)) : (hi >> 8)));
    return v & 0xFFFFFFu;
}
// taxon of a slot as the aggregators want it: 0 = miss (what `prot2kmer2lca -o` prints, src/commands/prot2kmer2lca.rs:176-178)
__device__ __forceinline__ uint32_t load_value(const uint32_t* __restrict__ ret, uint32_t slot, uint32_t value_bytes) {
    const uint32_t v = load_raw(ret, slot, value_bytes);
    return v == miss_raw(value_bytes) ? 0u : v;
}

}  // namespace wire
}  // namespace umgap
