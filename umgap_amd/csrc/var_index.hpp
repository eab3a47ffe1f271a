// var_index.hpp — what the two formats of the variable-length (tryptic peptide) index share: the handle, the legacy slot
// helpers (index_var.hip: 16-byte slots + key bytes in a heap), the staging of a batch of query bytes in LDS, and the layout of
// the PACKED format (index_varp.hip, round 3: the key itself in 1-3 adjacent slots, no heap).  gfx950 library internal.
#pragma once
#include "probe_common.hpp"

using umgap::probe::u32x4;

namespace umgap {
static constexpr uint32_t VAR_SLOTS = 4, VAR_MAX_DISP = 15, VAR_MAX_LEN = 63;
static constexpr uint64_t VAR_EMPTY = ~0ull;

static constexpr uint32_t VAR_INLINE_MAX = 17;
static constexpr uint64_t VAR_INLINE = 1ull << 63;

typedef uint64_t __attribute__((aligned(1))) u64_unaligned;

// (lo, hi) of an inline key: lo = the slot's offset word (flag included; never all ones: a residue code is < 26), hi = the
// 22 bits that go to the tag field.  `byte(i)` yields key byte i.  False: the key does not qualify.
template <class Byte>
__host__ __device__ inline bool pack_inline_with(Byte byte, uint32_t len, uint64_t& lo, uint32_t& hi) {
    lo = 0;
    hi = 0;
    if (len > VAR_INLINE_MAX) return false;
    uint64_t x0 = 0, x1 = 0;
    bool ok = true;
#pragma unroll
    for (uint32_t i = 0; i < VAR_INLINE_MAX; i++) {
        if (i < len) {
            const uint32_t c = (uint32_t)byte(i) - 'A';
            ok = ok && c < 26u;
            const uint32_t sh = 5 * i;
            if (sh < 64) {
                x0 |= (uint64_t)(c & 31u) << sh;
                if (sh > 59) x1 |= (uint64_t)(c & 31u) >> (64 - sh);
            } else x1 |= (uint64_t)(c & 31u) << (sh - 64);
        }
    }
    lo = x0 | VAR_INLINE;
    hi = (uint32_t)((x1 << 1) | (x0 >> 63));
    return ok;
}
__host__ __device__ inline bool pack_inline(const uint8_t* p, uint32_t len, uint64_t& lo, uint32_t& hi) {
    return pack_inline_with([p](uint32_t i) { return p[i]; }, len, lo, hi);
}
// The same from the key's first 20 bytes as five little-endian words (bytes past `len` may hold anything): four residues per
// word at a time -- the probe's version, ~80 VALU operations per key instead of ~170.
__device__ __forceinline__ bool pack_inline_words(const uint32_t (&d)[5], uint32_t len, uint64_t& lo, uint32_t& hi) {
    uint32_t u[5], bad = 0;
#pragma unroll
    for (int w = 0; w < 5; w++) {
        const int nv = (int)len - 4 * w;                                     // key bytes in this word (may be <= 0 or >= 4)
        const uint32_t mask = nv >= 4 ? 0xFFFFFFFFu : (nv <= 0 ? 0u : (0xFFFFFFFFu >> (32 - 8 * nv)));
        const uint32_t x = d[w];
        // a byte outside 'A'..'Z' raises its top bit in one of the three terms (a borrow or carry only travels upwards, into
        // bytes that are then bad or masked themselves)
        bad |= (x | (x + 0x25252525u) | ~((x | 0x80808080u) - 0x41414141u)) & 0x80808080u & mask;
        const uint32_t c = (x - 0x41414141u) & mask;                          // four residue codes, 8 bits apart
        const uint32_t t = (c & 0x001F001Fu) | ((c & 0x1F001F00u) >> 3);      // 10 + 10 bits, 16 bits apart
        u[w] = (t & 0x3FFu) | ((t >> 6) & 0xFFC00u);                          // 20 bits
    }
    const uint64_t x0 = (uint64_t)u[0] | ((uint64_t)u[1] << 20) | ((uint64_t)u[2] << 40) | ((uint64_t)u[3] << 60);
    const uint32_t x1 = (u[3] >> 4) | (u[4] << 16);
    lo = x0 | VAR_INLINE;
    hi = (x1 << 1) | (uint32_t)(x0 >> 63);
    return bad == 0;
}
// the key bytes of an inline slot (word0 = tag field | len << 25 | ..., off = offset word); returns the length
__host__ inline uint32_t unpack_inline(uint32_t word0, uint64_t off, uint8_t* out) {
    const uint32_t len = (word0 >> 25) & 63u, hi = word0 & 0x01FFFFFFu;
    const uint64_t x0 = (off & ~VAR_INLINE) | ((uint64_t)(hi & 1u) << 63), x1 = hi >> 1;
    for (uint32_t i = 0; i < len && i < VAR_INLINE_MAX; i++) {
        const uint32_t sh = 5 * i;
        uint64_t c = sh < 64 ? (x0 >> sh) | (sh > 59 ? x1 << (64 - sh) : 0ull) : x1 >> (sh - 64);
        out[i] = (uint8_t)('A' + (c & 31u));
    }
    return len;
}

// 8 bytes at a time (unaligned 64-bit loads; the last word overlaps the previous one when len is
// not a multiple of 8), byte-wise only for keys shorter than 8
__host__ __device__ inline uint64_t hash_bytes(const uint8_t* p, uint32_t len) {
    uint64_t h = 0xCBF29CE484222325ull ^ ((uint64_t)len * 0x9E3779B97F4A7C15ull);
    if (len >= 8) {
        uint32_t i = 0;
        for (; i + 8 <= len; i += 8) {
            h = (h ^ *reinterpret_cast<const u64_unaligned*>(p + i)) * 0xD6E8FEB86659FD93ull;
            h ^= h >> 29;
        }
        if (i < len) {
            h = (h ^ *reinterpret_cast<const u64_unaligned*>(p + len - 8)) * 0xD6E8FEB86659FD93ull;
            h ^= h >> 29;
        }
    } else {
        for (uint32_t i = 0; i < len; i++) h = (h ^ p[i]) * 0x100000001B3ull;
    }
    h ^= h >> 32;
    h *= 0xD6E8FEB86659FD93ull;
    h ^= h >> 32;
    return h;
}
}  // namespace umgap


// ---------------------------------------------------------------------------------------------------------------------
// PACKED format (round 3).  Tryptic peptides are words over 'A'..'Z' of 5..50 letters; at 5 bits per residue even a 63-letter key
// is 315 bits, so the key can live IN the table: an entry takes S = 1, 2 or 3 adjacent 16-byte slots of ONE 64-byte bucket
//     header slot        x = taxon   y = len (6 bits) | key bits   z, w = key bits,  w bit 31 = 1
//     continuation slot  x, y, z, w = key bits,  w bit 31 = 0
// (16 residues in the header, 24 per continuation: S = 1 up to 16 letters, 2 up to 40, 3 up to 63), entries packed from slot 0 up.
// A probe reads the home bucket -- one request -- and compares whole slots with the query's packed images: the header image
// must equal a header slot (the marker bit keeps continuations out) and the next S - 1 slots the continuation images.  Exact, hit or
// miss, with no second request: the heap, its 29.5 GB per G keys and its one-to-two sectors per hit are gone.
// A key that did not fit its home bucket b sits in bucket b + d, 1 <= d <= VARP_MAX_DISP, or in the stash (packed keys, sorted by
// hash); slot 0 of b -- occupied whenever that happens -- records WHICH: bit 22 + d of its w for displacement d, bit 30 for the
// stash.  So the home bucket tells a probe exactly where else the key can be: a miss ends at the home bucket unless a key HOMED
// there was displaced (keys merely passing through leave no mark, unlike the legacy format's per-bucket overflow flag), and the
// other buckets of all 64 queries of a batch are fetched together, one more round trip whatever the displacement.
// A table holds either format; keys with a byte outside 'A'..'Z' keep the legacy one (var_build decides).
// ---------------------------------------------------------------------------------------------------------------------
namespace umgap {
static constexpr uint32_t VARP_HDR = 16, VARP_CONT = 24;          // residues per header / continuation slot
static constexpr uint32_t VARP_MARK = 0x80000000u;                // w of a header slot
static constexpr uint32_t VARP_MAX_DISP = 7;                      // <= VAR_MAX_DISP (the table's halo)
static constexpr uint32_t VARP_DISP_SHIFT = 22;                   // w of slot 0: bit 22 + d = a key homed here sits d buckets on
static constexpr uint32_t VARP_STASHED = 1u << 30;                // w of slot 0: a key homed here is in the stash
static constexpr uint32_t VARP_WHERE = 0x7F800000u;               // those eight bits
static constexpr uint32_t VARP_INLINE_ALL = 64;                   // VarView::inline_max / file header value that names this format

__host__ __device__ inline uint32_t varp_slots(uint32_t len) { return len <= VARP_HDR ? 1u : (len <= VARP_HDR + VARP_CONT ? 2u : 3u); }

// a key as the table holds it: q[0] = header image (x = 0: the taxon is not part of the key), q[1], q[2] = continuations (zero when unused)
struct VarpKey {
    uint32_t q[3][4];
};

// images from the 16 groups g[i] = residues 4i .. 4i+3 at 5 bits each (zero past the key's end)
__host__ __device__ inline void varp_images(const uint32_t (&g)[16], uint32_t len, VarpKey& K) {
    K.q[0][0] = 0u;
    K.q[0][1] = len | (g[0] << 7) | (g[1] << 27);
    K.q[0][2] = (g[1] >> 5) | (g[2] << 15);
    K.q[0][3] = (g[2] >> 17) | (g[3] << 3) | VARP_MARK;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < 2; k++) {
        const uint32_t* a = g + 4 + 6 * k;
        K.q[1 + k][0] = a[0] | (a[1] << 20);
        K.q[1 + k][1] = (a[1] >> 12) | (a[2] << 8) | (a[3] << 28);
        K.q[1 + k][2] = (a[3] >> 4) | (a[4] << 16);
        K.q[1 + k][3] = (a[4] >> 16) | (a[5] << 4);
    }
}
// from the key bytes (host, build kernel, the probe's global-memory path); false: a byte outside 'A'..'Z' or a bad length
__host__ __device__ inline bool varp_pack_bytes(const uint8_t* p, uint32_t len, VarpKey& K) {
    uint32_t g[16];
    bool ok = len >= 1 && len <= VAR_MAX_LEN;
    for (uint32_t i = 0; i < 16; i++) {
        uint32_t v = 0;
        for (uint32_t t = 0; t < 4; t++) {
            const uint32_t r = 4 * i + t;
            if (ok && r < len) {
                const uint32_t c = (uint32_t)p[r] - 'A';
                if (c >= 26u) ok = false;
                v |= (c & 31u) << (5 * t);
            }
        }
        g[i] = v;
    }
    if (!ok) {
        for (uint32_t i = 0; i < 16; i++) g[i] = 0;
        len = 0;
    }
    varp_images(g, len, K);
    if (!ok) K.q[0][3] = 0u;      // all-zero images: "no key"
    return ok;
}
// the key bytes back (printindex); returns the length
__host__ inline uint32_t varp_unpack(const uint32_t* hdr, const uint32_t* c1, const uint32_t* c2, uint8_t* out) {
    const uint32_t len = hdr[1] & 63u;
    uint32_t g[16] = {0};
    g[0] = (hdr[1] >> 7) & 0xFFFFFu;
    g[1] = ((hdr[1] >> 27) | (hdr[2] << 5)) & 0xFFFFFu;
    g[2] = ((hdr[2] >> 15) | (hdr[3] << 17)) & 0xFFFFFu;
    g[3] = (hdr[3] >> 3) & 0xFFFFFu;
    const uint32_t* cs[2] = {c1, c2};
    for (int k = 0; k < 2; k++) {
        const uint32_t* c = cs[k];
        if (!c) continue;
        uint32_t* a = g + 4 + 6 * k;
        a[0] = c[0] & 0xFFFFFu;
        a[1] = ((c[0] >> 20) | (c[1] << 12)) & 0xFFFFFu;
        a[2] = (c[1] >> 8) & 0xFFFFFu;
        a[3] = ((c[1] >> 28) | (c[2] << 4)) & 0xFFFFFu;
        a[4] = ((c[2] >> 16) | (c[3] << 16)) & 0xFFFFFu;
        a[5] = (c[3] >> 4) & 0xFFFFFu;
    }
    for (uint32_t r = 0; r < len; r++) out[r] = (uint8_t)('A' + ((g[r >> 2] >> (5 * (r & 3))) & 31u));
    return len;
}
// hash of a packed key (the images that are in use; the length is part of the header image).  32 bits, one 32-bit multiplication per
// two words: 64-bit multiplications are a dozen quarter-rate instructions each on this chip and the probe computes one hash per lookup.
// The home bucket is the high half of hash x buckets (the table has fewer than 2^32 buckets).
__host__ __device__ inline uint32_t varp_hash(const VarpKey& K) {
    const uint32_t S = varp_slots(K.q[0][1] & 63u), C1 = 0x85EBCA6Bu, C2 = 0xC2B2AE35u;
    // two words per multiplication: the second one enters rotated into the product
    auto rot = [](uint32_t v, int r) { return (v << r) | (v >> (32 - r)); };
    uint32_t a = (K.q[0][1] ^ 0x9E3779B9u) * C1;
    a = ((a ^ (a >> 15)) ^ K.q[0][2]) * C2 + rot(K.q[0][3], 13);
    if (S >= 2) {
        a = ((a ^ (a >> 13)) ^ K.q[1][0]) * C1 + rot(K.q[1][1], 7);
        a = ((a ^ (a >> 16)) ^ K.q[1][2]) * C2 + rot(K.q[1][3], 19);
    }
    if (S >= 3) {
        a = ((a ^ (a >> 15)) ^ K.q[2][0]) * C1 + rot(K.q[2][1], 11);
        a = ((a ^ (a >> 13)) ^ K.q[2][2]) * C2 + rot(K.q[2][3], 23);
    }
    a ^= a >> 16;
    a *= C1;
    a ^= a >> 13;
    a *= C2;
    a ^= a >> 16;
    return a;
}
__host__ __device__ inline uint64_t varp_home(uint32_t hash, uint64_t nb) { return ((uint64_t)hash * (uint64_t)(uint32_t)nb) >> 32; }

// stash of a packed table: keys whose displacement chain ran out, sorted by hash
struct VarpStashEntry {
    uint64_t hash;        // varp_hash (32 bits)
    uint32_t img[11];     // header y, z, w; continuation 1 x..w; continuation 2 x..w
    uint32_t taxon;
};
}  // namespace umgap

struct VarView {
    const u32x4* slots;     // nb + VAR_MAX_DISP buckets x 4 slots
    const uint8_t* heap;
    uint64_t nb;
    const uint64_t* stash_hash;  // sorted
    const uint64_t* stash_off;
    const uint32_t* stash_val;
    uint32_t n_stash;
    uint32_t inline_max;    // keys of at most this many letters live in their slot (0: none do -- tables written before round 2;
                            // VARP_INLINE_ALL: the packed format, every key in 1-3 slots, no heap)
    const umgap::VarpStashEntry* pstash;   // packed format: its stash (n_stash entries)
};

struct umgap_var_index {
    u32x4* d_slots = nullptr;
    uint8_t* d_heap = nullptr;
    uint64_t heap_bytes = 0, nb = 0, n_keys = 0;
    uint64_t* d_stash_hash = nullptr;
    uint64_t* d_stash_off = nullptr;
    uint32_t* d_stash_val = nullptr;
    uint32_t n_stash = 0;
    uint32_t inline_max = 0;
    umgap::VarpStashEntry* d_pstash = nullptr;
    bool packed() const { return inline_max == umgap::VARP_INLINE_ALL; }
    VarView view() const { return VarView{d_slots, d_heap, nb, d_stash_hash, d_stash_off, d_stash_val, n_stash, inline_max, d_pstash}; }
};

// ---- staging of a batch of query bytes in the wave's LDS strip (both formats) ----
namespace umgap {
constexpr uint32_t VAR_STAGE = 64 * VAR_MAX_LEN + 16 + 48;   // a batch's bytes + alignment slack + read-ahead of the dword window
typedef u32x4 __attribute__((aligned(1))) u32x4_unaligned;

// 32 / 64 bits of the staged bytes at byte offset s (any alignment)
__device__ __forceinline__ uint32_t lds_load32(const uint32_t* w, uint32_t s) {
    const uint32_t k = s >> 2;
    return __builtin_amdgcn_alignbyte(w[k + 1], w[k], s & 3u);
}
__device__ __forceinline__ uint64_t lds_load64(const uint32_t* w, uint32_t s) {
    const uint32_t k = s >> 2, sh = s & 3u;
    const uint32_t a = w[k], b = w[k + 1], c = w[k + 2];
    return ((uint64_t)__builtin_amdgcn_alignbyte(c, b, sh) << 32) | __builtin_amdgcn_alignbyte(b, a, sh);
}
// Where a batch's bytes go: the byte range [first, end) of the query buffer that holds its keys, from the 16-byte aligned address a0
// on.  ok = every key lies inside it, it fits the strip and the buffer -- otherwise the batch takes the global-memory path.
struct VarStage {
    uint64_t a0, span;
    bool ok;
};
// o / o1: the lane's key is bytes [o, o1) of the buffer (lanes past the end of the list hold an empty key); first / end: the range
// (min_len / max_len: 1 <= min_len, max_len <= VAR_MAX_LEN; a key whose length lies outside is answered with a miss without a probe --
// the length filter of prot2tryp2lca, src/commands/prot2tryp2lca.rs:117-120)
__device__ __forceinline__ VarStage var_stage_plan(const uint8_t* qbytes, uint64_t all_end, uint64_t o, uint64_t o1, uint64_t first, uint64_t end,
                                                   uint32_t force_global, uint32_t min_len, uint32_t max_len, uint32_t& meta) {
    const uint64_t l64 = o1 - o;
    const bool valid = o1 >= o && l64 >= min_len && l64 <= max_len;
    const uint32_t len = valid ? (uint32_t)l64 : 0u;
    const bool inside = !valid || (o >= first && o + len <= end);
    VarStage S;
    S.a0 = (uint64_t)(uintptr_t)(qbytes + first) & ~15ull;                                    // absolute, 16-byte aligned
    S.span = end >= first ? (uint64_t)(uintptr_t)(qbytes + end) - S.a0 : ~0ull;
    S.ok = S.span <= 64 * VAR_MAX_LEN + 16 && end <= all_end && !__any(!inside) && !force_global;
    const uint32_t soff = S.ok ? (uint32_t)((uint64_t)(uintptr_t)(qbytes + o) - S.a0) : 0u;
    meta = (soff & 0x1FFFu) | (len << 13) | ((valid && S.ok ? 1u : 0u) << 19);
    return S;
}
// one 16-byte chunk that straddles an end of the query buffer: byte by byte, inside only
static __device__ __noinline__ u32x4 var_edge_chunk(uint64_t a, uint64_t lo_abs, uint64_t hi_abs) {
    uint32_t wds[4] = {0u, 0u, 0u, 0u};
    for (uint32_t t = 0; t < 16; t++)
        if (a + t >= lo_abs && a + t < hi_abs) wds[t >> 2] |= (uint32_t)*reinterpret_cast<const uint8_t*>(a + t) << (8 * (t & 3));
    return u32x4{wds[0], wds[1], wds[2], wds[3]};
}
}  // namespace umgap
