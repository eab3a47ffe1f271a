// tryptic.hip — the front half of `prot2tryp2lca` on the device, and the tryptic pipeline fused behind it (gfx950 only).
//
// Reference (src/commands/prot2tryp2lca.rs:107-131): every sequence of a record is cut with the pattern `([KR])([^P])` ->
// "$1\n$2" applied twice (a letter can end one match and start the next), `*` becomes a newline, empty lines go, peptides
// outside [minlen, maxlen] go, the rest is looked up.  With that (default) pattern the cut is a per-byte rule: a new piece
// starts at byte p when p opens a fragment, when bytes[p-1] is K or R and bytes[p] is not P, or at and after a `*`.
// Every byte then belongs to exactly one piece and the pieces of a fragment lie back to back, so the digest is just the sorted
// list of piece starts -- which is the (bytes, offsets) key list the variable-length probe takes as it is: no byte moves.
// A `*` is a piece of its own here; it and the pieces outside the length window are answered with a miss without a probe
// (the window goes into the probe kernel, index_var.hip), and 0 = "no hit" is what the aggregators skip.
//
//   k_tryp_mark     fragment starts -> one bit per byte                                     (n_fragments threads)
//   k_tryp_count    piece starts per tile of 8 KB, the flags kept (1 bit per byte)            (reads the bytes, once)
//   k_tryp_scan     exclusive scan of the tile counts, two levels                             (a workgroup per 16384 tiles + one)
//   k_tryp_write    the starts, in order: offsets[rank] = position                           (reads the flags)
//   k_tryp_reads    first piece of every read = the rank of its first byte among the starts   (n_reads threads, three loads each)
// then k_lookup_var_pipe over the pieces and the CSR aggregate over the hits of each read.
#include <algorithm>
#include <vector>

#include "common.hpp"
#include "taxonomy.hpp"

namespace umgap {
int use_device(int device);
int var_lookup_dev(const umgap_var_index* v, const uint8_t* d_bytes, const uint64_t* d_off, size_t n, uint32_t miss, uint32_t* d_out, hipStream_t s,
                   uint32_t min_len, uint32_t max_len);
int aggregate_csr_bound_dev(umgap_taxonomy* tx, int method, int strategy, float factor, float lower_bound, int ranked_only, uint32_t* d_codes,
                            uint64_t n_hits, const uint64_t* d_read_offsets, size_t n_reads, uint32_t* d_out_taxon, uint32_t* d_out_status, hipStream_t stream);
int aggregate_csr_counted_dev(umgap_taxonomy* tx, int method, int strategy, float factor, float lower_bound, int ranked_only, const uint32_t* d_taxa,
                              uint64_t n_hits, const uint64_t* d_read_offsets, size_t n_reads, uint32_t* d_out_taxon, uint32_t* d_out_status, hipStream_t stream);
}  // namespace umgap

using namespace umgap;

namespace {

constexpr uint32_t PER_THREAD = 32, TILE = 256 * PER_THREAD;   // bytes per thread / per workgroup of 256 threads (8 KB)

__global__ __launch_bounds__(256) void k_tryp_mark(const uint64_t* __restrict__ foff, size_t n_frag, uint64_t b0, uint64_t b1, uint32_t* __restrict__ bits,
                                                   unsigned long long* __restrict__ bad) {
    size_t f = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (; f < n_frag; f += step) {
        const uint64_t o = foff[f], o1 = foff[f + 1];
        if (o1 < o || o < b0 || o1 > b1) { atomicAdd(bad, 1ull); continue; }
        if (o1 > o) atomicOr(bits + ((o - b0) >> 5), 1u << ((o - b0) & 31u));
    }
}

// 0x80 in every byte of v that equals c (exact: no carry leaves a byte)
__device__ __forceinline__ uint64_t bytes_equal_to(uint64_t v, uint8_t c) {
    const uint64_t x = v ^ (0x0101010101010101ull * c), low7 = 0x7F7F7F7F7F7F7F7Full;
    return ~(((x & low7) + low7) | x | low7);
}

// the piece-start flags of the 32 bytes [rel, rel + 32) of the range (bit k: byte rel + k); n = bytes in the range.  Eight bytes
// at a time: K / R / P / `*` are found with byte-parallel arithmetic on the 64-bit word, "the byte before" is a shift by one byte
// with the previous word's last byte carried in, and the eight flags are gathered with one multiplication.
__device__ __forceinline__ uint32_t piece_start_mask(const uint8_t* __restrict__ base, const uint32_t* __restrict__ bits, uint64_t rel, uint64_t n) {
    if (rel >= n) return 0u;
    typedef uint64_t __attribute__((aligned(1))) u64_unaligned;
    uint8_t prev = rel ? base[rel - 1] : (uint8_t)'*';         // before the range's first byte: any separator (that byte opens a fragment anyway)
    uint32_t m = bits[rel >> 5];                                // rel is a multiple of 32: the 32 fragment-start bits are one word
#pragma unroll
    for (uint32_t w = 0; w < PER_THREAD / 8; w++) {
        const uint64_t at = rel + 8 * w;
        if (at >= n) break;
        uint64_t v = 0;
        if (at + 8 <= n) v = *reinterpret_cast<const u64_unaligned*>(base + at);
        else for (uint32_t k = 0; at + k < n; k++) v |= (uint64_t)base[at + k] << (8 * k);
        const uint64_t star = bytes_equal_to(v, '*'), kr = bytes_equal_to(v, 'K') | bytes_equal_to(v, 'R'), isp = bytes_equal_to(v, 'P');
        const uint64_t prev_star = (star << 8) | (prev == '*' ? 0x80ull : 0ull), prev_kr = (kr << 8) | ((prev == 'K' || prev == 'R') ? 0x80ull : 0ull);
        const uint64_t start = star | prev_star | (prev_kr & ~isp);                       // 0x80 per byte that opens a piece
        m |= (uint32_t)(((start >> 7) * 0x0102040810204080ull) >> 56) << (8 * w);         // bit k <- byte k
        prev = (uint8_t)(v >> 56);
    }
    const uint64_t left = n - rel;                               // bytes past the end open nothing
    return left >= 32 ? m : (m & ((1u << (uint32_t)left) - 1u));
}

__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* lds_wave_sums, uint32_t& block_total) {
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)incl, o, 64); if (lane >= (uint32_t)o) incl += t; }
    if (lane == 63) lds_wave_sums[wv] = incl;
    __syncthreads();
    uint32_t before = 0, total = 0;
    for (uint32_t w = 0; w < blockDim.x / 64; w++) { const uint32_t s = lds_wave_sums[w]; if (w < wv) before += s; total += s; }
    __syncthreads();
    block_total = total;
    return before + incl - v;
}

// the piece-start flags of 16 bytes held in two little-endian words (lo = bytes 0..7), `prev` = the byte before them: 16 bits
__device__ __forceinline__ uint32_t piece_start_mask16(uint64_t lo, uint64_t hi, uint8_t prev) {
    uint32_t m = 0;
    const uint64_t v2[2] = {lo, hi};
#pragma unroll
    for (int w = 0; w < 2; w++) {
        const uint64_t v = v2[w];
        const uint64_t star = bytes_equal_to(v, '*'), kr = bytes_equal_to(v, 'K') | bytes_equal_to(v, 'R'), isp = bytes_equal_to(v, 'P');
        const uint64_t prev_star = (star << 8) | (prev == '*' ? 0x80ull : 0ull), prev_kr = (kr << 8) | ((prev == 'K' || prev == 'R') ? 0x80ull : 0ull);
        const uint64_t start = star | prev_star | (prev_kr & ~isp);
        m |= (uint32_t)(((start >> 7) * 0x0102040810204080ull) >> 56) << (8 * w);
        prev = (uint8_t)(v >> 56);
    }
    return m;
}

// Piece starts per tile of 8 KB.  Also keeps the 32 flags of every 32-byte group (one bit per byte of input: k_tryp_write reads those
// instead of the bytes again) and how many pieces start in the tile before the group (16 bits: a tile has 8192 bytes): with the tile
// bases that is the rank of any byte among the piece starts in three loads (k_tryp_reads).
// The bytes are read 16 per lane, a wave a whole KB per instruction (round 4; 32 bytes per thread as four 8-byte loads left three
// quarters of every request unused: 3.3 TB/s); the byte before a lane's 16 comes from the lane below, two lanes make a group, and
// the groups' flags meet in LDS for the tile's scan.
constexpr uint32_t TILES_PER_GROUP = 1;      // tiles a workgroup takes in a row (4 measured slower: 424 against 373 us per 1.34 GB)
__global__ __launch_bounds__(256) void k_tryp_count(const uint8_t* __restrict__ base, const uint32_t* __restrict__ bits, uint64_t n, size_t n_tiles,
                                                    uint32_t* __restrict__ tile_counts, uint32_t* __restrict__ masks, uint16_t* __restrict__ before) {
    __shared__ uint32_t sums[4];
    __shared__ uint32_t gmask[256];
    const uint32_t t = threadIdx.x, lane = t & 63;
    for (size_t tile = (size_t)blockIdx.x * TILES_PER_GROUP; tile < n_tiles && tile < ((size_t)blockIdx.x + 1) * TILES_PER_GROUP; tile++) {
    const uint64_t tile0 = (uint64_t)tile * TILE;
    typedef uint64_t __attribute__((ext_vector_type(2))) u64x2;
    typedef u64x2 __attribute__((aligned(1))) u64x2_unaligned;        // (the range starts wherever the first fragment does)
#pragma unroll
    for (uint32_t r = 0; r < 2; r++) {
        const uint64_t rel = tile0 + (uint64_t)(r * 256 + t) * 16;
        uint64_t lo = 0, hi = 0;
        if (rel + 16 <= n) {
            const u64x2 v = *reinterpret_cast<const u64x2_unaligned*>(base + rel);
            lo = v.x; hi = v.y;
        } else if (rel < n) {                                          // the tail of the range
            for (uint32_t k = 0; k < 16 && rel + k < n; k++) {
                const uint64_t b = base[rel + k];
                if (k < 8) lo |= b << (8 * k); else hi |= b << (8 * (k - 8));
            }
        }
        // the byte before the lane's 16: the last byte of the lane below; lane 0 of a wave reads it (before the range's first byte:
        // any separator -- that byte opens a fragment anyway)
        uint32_t prev = (uint32_t)__shfl_up((int)(uint32_t)(hi >> 56), 1, 64);
        if (lane == 0) prev = rel && rel <= n ? base[rel - 1] : (uint32_t)'*';
        uint32_t m16 = rel < n ? piece_start_mask16(lo, hi, (uint8_t)prev) : 0u;
        if (rel < n && n - rel < 16) m16 &= (1u << (uint32_t)(n - rel)) - 1u;      // bytes past the end open nothing
        const uint32_t up = (uint32_t)__shfl_down((int)m16, 1, 64);               // the odd lane's half of the group
        if (!(t & 1u)) gmask[r * 128 + (t >> 1)] = m16 | (up << 16);
    }
    __syncthreads();
    const uint64_t grel = tile0 + (uint64_t)t * PER_THREAD;           // thread t: group t of the tile
    const uint32_t mask = grel < n ? (gmask[t] | bits[grel >> 5]) : 0u;      // (+ the fragment-start bits of the group: one word)
    masks[(uint64_t)tile * 256 + t] = mask;
    const uint32_t cnt = (uint32_t)__builtin_popcount(mask);
    uint32_t total;
    before[(uint64_t)tile * 256 + t] = (uint16_t)block_exclusive_scan(cnt, sums, total);     // (ends with a barrier: gmask is free again)
    if (t == 0) tile_counts[tile] = total;
    }
}

// Exclusive scan of the n_tiles counts in two levels (round 4; one workgroup walking the whole list was held to one CU's memory
// bandwidth: 0.11 ms for the 160 K tiles of a 10 M-pair batch).  Level 1: a workgroup of 1024 threads per SLAB of 16384 tiles (sixteen
// consecutive counts per thread as four 16-byte loads) -> tile_base[i] = pieces of the slab before tile i, slab_total[b].
// Level 2: one workgroup scans the slab totals -> slab_base[b], *total.  A tile's base is tile_base[i] + slab_base[i / SLAB].
constexpr uint32_t SCAN_PER = 16, SLAB = 1024 * SCAN_PER;
__global__ __launch_bounds__(1024) void k_tryp_scan(const uint32_t* __restrict__ tile_counts, size_t n_tiles, uint32_t* __restrict__ tile_base, uint64_t* __restrict__ slab_total) {
    __shared__ uint32_t sums[16];
    const size_t i = (size_t)blockIdx.x * SLAB + (size_t)threadIdx.x * SCAN_PER;
    uint32_t v[SCAN_PER], mine = 0;
    if (i + SCAN_PER <= n_tiles) {               // (the scratch is carved at 256-byte boundaries: 16-byte loads are aligned)
#pragma unroll
        for (uint32_t k = 0; k < SCAN_PER; k += 4) {
            const uint4 q = *reinterpret_cast<const uint4*>(tile_counts + i + k);
            v[k] = q.x; v[k + 1] = q.y; v[k + 2] = q.z; v[k + 3] = q.w;
        }
    } else {
#pragma unroll
        for (uint32_t k = 0; k < SCAN_PER; k++) v[k] = i + k < n_tiles ? tile_counts[i + k] : 0u;
    }
#pragma unroll
    for (uint32_t k = 0; k < SCAN_PER; k++) mine += v[k];
    uint32_t total;                              // (a slab holds 16384 tiles x 8192 bytes: its pieces fit 32 bits)
    uint32_t at = block_exclusive_scan(mine, sums, total);
#pragma unroll
    for (uint32_t k = 0; k < SCAN_PER; k++) { if (i + k < n_tiles) tile_base[i + k] = at; at += v[k]; }
    if (threadIdx.x == 0) slab_total[blockIdx.x] = total;
}
// (report: page-locked host words, or null -- [2] = the number of pieces, [3] = the fragments k_tryp_mark refused)
__global__ __launch_bounds__(1024) void k_tryp_scan_slabs(const uint64_t* __restrict__ slab_total, size_t n_slabs, uint64_t* __restrict__ slab_base, uint64_t* __restrict__ total,
                                                          const unsigned long long* __restrict__ bad, uint64_t* __restrict__ report) {
    __shared__ uint64_t wsum[16];
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint64_t run = 0;
    for (size_t s0 = 0; s0 < n_slabs; s0 += 1024) {
        const size_t i = s0 + threadIdx.x;
        const uint64_t v = i < n_slabs ? slab_total[i] : 0ull;
        uint64_t incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint64_t t = ((uint64_t)(uint32_t)__shfl_up((int)(uint32_t)(incl >> 32), o, 64) << 32) | (uint32_t)__shfl_up((int)(uint32_t)incl, o, 64);
            if (lane >= (uint32_t)o) incl += t;
        }
        if (lane == 63) wsum[wv] = incl;
        __syncthreads();
        uint64_t before = 0, all = 0;
        for (uint32_t w = 0; w < 16; w++) { const uint64_t x = wsum[w]; if (w < wv) before += x; all += x; }
        __syncthreads();
        if (i < n_slabs) slab_base[i] = run + before + incl - v;
        run += all;
    }
    if (threadIdx.x == 0) {
        *total = run;
        if (report) { report[2] = run; report[3] = *bad; }
    }
}
// the first and the last fragment offset, straight into page-locked host memory
__global__ void k_tryp_ends(const uint64_t* __restrict__ foff, size_t n_frag, uint64_t* __restrict__ report) {
    if (threadIdx.x == 0) { report[0] = foff[0]; report[1] = foff[n_frag]; }
}

__global__ __launch_bounds__(256) void k_tryp_write(const uint32_t* __restrict__ masks, const uint16_t* __restrict__ before, uint64_t n, uint64_t b0,
                                                    const uint32_t* __restrict__ tile_base, const uint64_t* __restrict__ slab_base, uint64_t total, uint64_t cap,
                                                    uint64_t* __restrict__ offsets) {
    // a thread per 32-byte group: its first piece is number slab base + tile base + the tile's starts before the group (kept by
    // k_tryp_count) -- no scan of its own, no barrier, any grid
    const uint64_t n_groups = (n + PER_THREAD - 1) / PER_THREAD;
    uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t step = (uint64_t)gridDim.x * blockDim.x;
    for (; g < n_groups; g += step) {
        uint32_t m = masks[g];
        if (!m) continue;
        const uint64_t tile = g / 256, rel = g * PER_THREAD;
        uint64_t at = slab_base[tile / SLAB] + tile_base[tile] + before[g];
        while (m) {
            const uint32_t k = (uint32_t)__builtin_ctz(m);
            m &= m - 1;
            if (at < cap) offsets[at] = b0 + rel + k;
            at++;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && total < cap + 1) offsets[total] = b0 + n;   // the end of the last piece
}

// first piece of read r = first piece that starts at or after the read's first byte (an empty read owns no piece) = the number of
// piece starts before that byte: tile base + starts before the byte's 32-byte group in the tile + starts below it in the group
// (the binary search over the 40 M offsets this replaces took 0.31 ms per 10 M reads, a quarter of the digest)
__global__ __launch_bounds__(256) void k_tryp_reads(const uint64_t* __restrict__ foff, size_t n_reads, uint32_t F, uint64_t b0, uint64_t n,
                                                    const uint32_t* __restrict__ tile_base, const uint64_t* __restrict__ slab_base, const uint16_t* __restrict__ before,
                                                    const uint32_t* __restrict__ masks,
                                                    uint64_t total, uint64_t* __restrict__ read_pieces) {
    size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (; r <= n_reads; r += step) {
        uint64_t at = total;
        if (r < n_reads) {
            const uint64_t target = foff[r * F];
            if (target <= b0) at = 0;
            else if (target - b0 < n) {
                const uint64_t rel = target - b0, g = rel >> 5;          // group of 32 bytes = one thread of k_tryp_count
                const uint64_t tile = rel / TILE;
                at = slab_base[tile / SLAB] + tile_base[tile] + before[g] + (uint64_t)__builtin_popcount(masks[g] & ((1u << (rel & 31u)) - 1u));
            }
        }
        read_pieces[r] = at;
    }
}

// a piece that is a `*` was never a peptide: only needed when the length window admits one-byte keys
__global__ __launch_bounds__(256) void k_tryp_drop_stars(const uint8_t* __restrict__ bytes, const uint64_t* __restrict__ offsets, uint64_t total, uint32_t miss, uint32_t* __restrict__ hits) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (; i < total; i += step)
        if (bytes[offsets[i]] == '*') hits[i] = miss;
}

inline unsigned grid_for(size_t n) {
    size_t g = (n + 255) / 256;
    return (unsigned)std::max<size_t>(1, std::min<size_t>(g, 256 * 16));
}

// digest scratch of one call: one allocation (the caller's own, or a handle's grow-only cache) carved into the five arrays
struct Digest {
    uint32_t* bits = nullptr;
    uint32_t* tile_counts = nullptr;
    uint32_t* tile_base = nullptr;       // pieces of the tile's slab before the tile
    uint64_t* slab_total = nullptr;
    uint64_t* slab_base = nullptr;
    size_t n_slabs = 0;
    uint64_t* d_total = nullptr;
    unsigned long long* d_bad = nullptr;
    uint32_t* masks = nullptr;
    uint16_t* before = nullptr;
    uint64_t b0 = 0, n = 0, total = 0;
    size_t n_tiles = 0;
    void* owned = nullptr;              // freed by release(); null when the memory belongs to a cache
    void** cache = nullptr;             // grow-only cache slot to use instead of a fresh allocation
    size_t* cache_bytes = nullptr;
    void release() {
        if (owned) (void)hipFree(owned);
        owned = nullptr;
    }
};

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// `bytes` of device memory: from the cache slot (grown if too small; the stream is drained first, the old block may be in use) or fresh
int scratch_get(void** cache, size_t* cache_bytes, size_t bytes, hipStream_t s, void** out, void** owned) {
    if (cache) {
        if (*cache_bytes < bytes) {
            UMGAP_HIP(hipStreamSynchronize(s));
            if (*cache) (void)hipFree(*cache);
            *cache = nullptr;
            *cache_bytes = 0;
            const size_t want = bytes + bytes / 8;
            if (hipMalloc(cache, want) != hipSuccess) return fail(UMGAP_ERR_NOMEM, "tryptic pipeline: hipMalloc of the scratch failed");
            *cache_bytes = want;
        }
        *out = *cache;
        return UMGAP_OK;
    }
    if (hipMalloc(out, bytes) != hipSuccess) return fail(UMGAP_ERR_NOMEM, "tryptic digest: hipMalloc of the scratch failed");
    *owned = *out;
    return UMGAP_OK;
}

// counts the pieces of fragments [0, n_frag) (host round trip for the count); leaves what k_tryp_write needs in D
// (known_ends: d_foff[0] and d_foff[n_frag] when the caller has them already -- saves the first of the two round trips)
// (report: four page-locked words the kernels write the ends and the counts to -- a round trip is then a launch and a wait, not a
//  staged copy into pageable memory; null: copies)
int digest_count(const uint8_t* d_bytes, const uint64_t* d_foff, size_t n_frag, hipStream_t s, Digest& D, const uint64_t* known_ends = nullptr,
                 uint64_t* report = nullptr) {
    uint64_t ends[2] = {0, 0};
    if (known_ends) { ends[0] = known_ends[0]; ends[1] = known_ends[1]; }
    else if (report) {
        hipLaunchKernelGGL(k_tryp_ends, dim3(1), dim3(64), 0, s, d_foff, n_frag, report);
        UMGAP_HIP(hipStreamSynchronize(s));
        ends[0] = report[0]; ends[1] = report[1];
    } else {
        UMGAP_HIP(hipMemcpyAsync(&ends[0], d_foff, 8, hipMemcpyDeviceToHost, s));
        UMGAP_HIP(hipMemcpyAsync(&ends[1], d_foff + n_frag, 8, hipMemcpyDeviceToHost, s));
        UMGAP_HIP(hipStreamSynchronize(s));
    }
    if (ends[1] < ends[0]) return fail(UMGAP_ERR_INVALID, "tryptic digest: fragment offsets must not decrease");
    D.b0 = ends[0];
    D.n = ends[1] - ends[0];
    D.n_tiles = (size_t)((D.n + TILE - 1) / TILE);
    if (D.n_tiles > 0x7FFFFFFFull) return fail(UMGAP_ERR_INVALID, "tryptic digest: more than 4 TB of fragments in one call");
    const size_t words = (size_t)(D.n / 32) + 2, tiles = std::max<size_t>(D.n_tiles, 1);
    D.n_slabs = (tiles + SLAB - 1) / SLAB;
    const size_t b_bits = align256(words * 4), b_cnt = align256(tiles * 4), b_base = align256(tiles * 4), b_masks = align256(tiles * 256 * 4), b_before = align256(tiles * 256 * 2),
                 b_slab = align256(D.n_slabs * 8);
    void* mem = nullptr;
    int rc = scratch_get(D.cache, D.cache_bytes, b_bits + b_cnt + b_base + b_masks + b_before + 2 * b_slab + 512, s, &mem, &D.owned);
    if (rc) return rc;
    char* p = (char*)mem;
    D.bits = (uint32_t*)p; p += b_bits;
    D.tile_counts = (uint32_t*)p; p += b_cnt;
    D.tile_base = (uint32_t*)p; p += b_base;
    D.slab_total = (uint64_t*)p; p += b_slab;
    D.slab_base = (uint64_t*)p; p += b_slab;
    D.masks = (uint32_t*)p; p += b_masks;
    D.before = (uint16_t*)p; p += b_before;
    D.d_total = (uint64_t*)p; p += 8;
    D.d_bad = (unsigned long long*)p;
    UMGAP_HIP(hipMemsetAsync(D.bits, 0, words * 4, s));
    UMGAP_HIP(hipMemsetAsync(D.d_total, 0, 16, s));      // ... and d_bad behind it
    hipLaunchKernelGGL(k_tryp_mark, dim3(grid_for(n_frag)), dim3(256), 0, s, d_foff, n_frag, D.b0, ends[1], D.bits, D.d_bad);
    if (D.n_tiles) {
        hipLaunchKernelGGL(k_tryp_count, dim3((unsigned)((D.n_tiles + TILES_PER_GROUP - 1) / TILES_PER_GROUP)), dim3(256), 0, s, d_bytes + D.b0, D.bits, D.n, D.n_tiles, D.tile_counts,
                           D.masks, D.before);
        hipLaunchKernelGGL(k_tryp_scan, dim3((unsigned)D.n_slabs), dim3(1024), 0, s, D.tile_counts, D.n_tiles, D.tile_base, D.slab_total);
        hipLaunchKernelGGL(k_tryp_scan_slabs, dim3(1), dim3(1024), 0, s, D.slab_total, D.n_slabs, D.slab_base, D.d_total, (const unsigned long long*)D.d_bad, report);
    }
    unsigned long long bad = 0;
    if (report && D.n_tiles) {
        UMGAP_HIP(hipStreamSynchronize(s));
        D.total = report[2];
        bad = report[3];
    } else {
        UMGAP_HIP(hipMemcpyAsync(&D.total, D.d_total, 8, hipMemcpyDeviceToHost, s));
        UMGAP_HIP(hipMemcpyAsync(&bad, D.d_bad, 8, hipMemcpyDeviceToHost, s));
        UMGAP_HIP(hipStreamSynchronize(s));
    }
    if (bad) return fail(UMGAP_ERR_INVALID, "tryptic digest: fragment offsets must not decrease");
    return UMGAP_OK;
}

// d_offsets holds cap + 1 entries, cap >= D.total (with no bytes at all the one block only writes the end: offsets[0] = b0)
int digest_write(hipStream_t s, const Digest& D, uint64_t* d_offsets, uint64_t cap) {
    hipLaunchKernelGGL(k_tryp_write, dim3(grid_for((size_t)((D.n + PER_THREAD - 1) / PER_THREAD))), dim3(256), 0, s, D.masks, D.before, D.n, D.b0, D.tile_base, D.slab_base, D.total,
                       cap, d_offsets);
    UMGAP_HIP(hipGetLastError());
    return UMGAP_OK;
}

}  // namespace

extern "C" {

int umgap_digest_tryp_dev(const uint8_t* d_bytes, const uint64_t* d_frag_offsets, size_t n_fragments, uint64_t* d_piece_offsets, size_t cap,
                          size_t* n_pieces, int device, void* stream) {
    if (!n_pieces || (n_fragments && (!d_bytes || !d_frag_offsets)) || (cap && !d_piece_offsets)) return fail(UMGAP_ERR_INVALID, "umgap_digest_tryp_dev: null argument");
    int rc = use_device(device);
    if (rc) return rc;
    *n_pieces = 0;
    if (!n_fragments) return UMGAP_OK;
    hipStream_t s = (hipStream_t)stream;
    Digest D;
    rc = digest_count(d_bytes, d_frag_offsets, n_fragments, s, D);
    if (!rc) {
        *n_pieces = (size_t)D.total;
        if (cap >= D.total + 1) {       // room for every start and the end
            rc = digest_write(s, D, d_piece_offsets, D.total);
            if (!rc && hipStreamSynchronize(s) != hipSuccess) rc = fail(UMGAP_ERR_DEVICE, "umgap_digest_tryp_dev: the digest did not complete");
        }
    }
    D.release();
    return rc;
}

int umgap_classify_tryp_dev(umgap_index* ix, umgap_taxonomy* tx, int method, int strategy, float factor, float lower_bound, int ranked_only,
                            const uint8_t* d_bytes, const uint64_t* d_frag_offsets, size_t n_reads, uint32_t fragments_per_read, uint32_t min_len,
                            uint32_t max_len, uint32_t* d_out_taxon, uint32_t* d_out_status, uint64_t* n_pieces_out, void* stream) {
    if (!ix || !tx || (n_reads && (!d_bytes || !d_frag_offsets || !d_out_taxon || !d_out_status))) return fail(UMGAP_ERR_INVALID, "umgap_classify_tryp_dev: null argument");
    if (ix->kind != 2 || !ix->var) return fail(UMGAP_ERR_INVALID, "umgap_classify_tryp_dev: needs a variable-length (tryptic) index");
    if (ix->device != tx->device) return fail(UMGAP_ERR_INVALID, "umgap_classify_tryp_dev: index and taxonomy live on different devices");
    // a table bound to THIS taxonomy (umgap_index_bind_taxonomy): the hits are interval codes and the aggregate gathers nothing by id
    if (ix->bound_to && ix->bound_to != tx) return fail(UMGAP_ERR_INVALID, "umgap_classify_tryp_dev: the index is bound to another taxonomy");
    if (!fragments_per_read) return fail(UMGAP_ERR_INVALID, "umgap_classify_tryp_dev: fragments_per_read must be at least 1");
    if (min_len < 1 || max_len < min_len) return fail(UMGAP_ERR_INVALID, "umgap_classify_tryp_dev: need 1 <= min_len <= max_len");
    int rc = use_device(ix->device);
    if (rc) return rc;
    if (n_pieces_out) *n_pieces_out = 0;
    if (!n_reads) return UMGAP_OK;
    hipStream_t s = (hipStream_t)stream;
    // eight page-locked words the digest's kernels report to (four per scratch pair): the call's two host round trips are then a launch
    // and a wait each, not staged copies into pageable memory
    if (!ix->tryp_host && hipHostMalloc((void**)&ix->tryp_host, 64, hipHostMallocDefault) != hipSuccess) {
        ix->tryp_host = nullptr;       // (without it the round trips are copies, as before: not an error)
        (void)hipGetLastError();
    }
    uint64_t* const report_words = getenv("UMGAP_TRYP_COPIES") ? nullptr : ix->tryp_host;     // (the switch is for A/B measurements)
    // One chunk of reads [r0, r1) on stream `ls` with the scratch pair `pair` of the handle: digest -> probe -> aggregate.  Everything
    // behind the count's host round trip is stream-ordered (grow-only scratch: no allocation in the steady state, nothing to free).
    // wait_agg / record_agg: the aggregate stages of the chunks of one call run one after another (the general kernels' large
    // scratch belongs to the taxonomy, not to a stream).
    auto run_chunk = [&](size_t r0, size_t r1, hipStream_t ls, int pair, const uint64_t* known_ends, hipEvent_t wait_agg, hipEvent_t record_agg) -> int {
        const size_t nr = r1 - r0, n_frag = nr * fragments_per_read;
        const uint64_t* foff = d_frag_offsets + r0 * fragments_per_read;
        Digest D;
        D.cache = &ix->tryp_scratch[2 * pair];
        D.cache_bytes = &ix->tryp_scratch_bytes[2 * pair];
        int rc = digest_count(d_bytes, foff, n_frag, ls, D, known_ends, report_words ? report_words + 4 * pair : nullptr);
        if (rc) return rc;
        if (n_pieces_out) *n_pieces_out += D.total;
        const size_t b_off = align256((D.total + 1) * 8), b_rp = align256((nr + 1) * 8), b_hits = align256(std::max<uint64_t>(D.total, 1) * 4);
        void *mem = nullptr, *unused = nullptr;
        rc = scratch_get(&ix->tryp_scratch[2 * pair + 1], &ix->tryp_scratch_bytes[2 * pair + 1], b_off + b_rp + b_hits, ls, &mem, &unused);
        if (rc) return rc;
        char* p = (char*)mem;
        uint64_t* d_offsets = (uint64_t*)p; p += b_off;
        uint64_t* d_read_pieces = (uint64_t*)p; p += b_rp;
        uint32_t* d_hits = (uint32_t*)p;
        rc = digest_write(ls, D, d_offsets, D.total);
        if (rc) return rc;
        // (k_tryp_reads needs the count pass only and is needed by the aggregate only; on a stream of its own beside the piece write and the
        //  probe it measured no gain -- 2.82-2.90 against 2.80 ms per call, same box -- and stays in line)
        hipLaunchKernelGGL(k_tryp_reads, dim3(grid_for(nr + 1)), dim3(256), 0, ls, foff, nr, fragments_per_read, D.b0, D.n, D.tile_base, D.slab_base, D.before, D.masks, D.total, d_read_pieces);
        if (D.total) {
            rc = var_lookup_dev(ix->var, d_bytes, d_offsets, (size_t)D.total, 0u, d_hits, ls, min_len, max_len);
            if (rc) return rc;
            if (min_len <= 1) hipLaunchKernelGGL(k_tryp_drop_stars, dim3(grid_for((size_t)D.total)), dim3(256), 0, ls, d_bytes, d_offsets, D.total, 0u, d_hits);
        }
        if (hipGetLastError() != hipSuccess) return fail(UMGAP_ERR_DEVICE, "umgap_classify_tryp_dev: a launch failed");
        if (wait_agg) UMGAP_HIP(hipStreamWaitEvent(ls, wait_agg, 0));
        if (ix->bound_to) rc = aggregate_csr_bound_dev(tx, method, strategy, factor, lower_bound, ranked_only, d_hits, D.total, d_read_pieces, nr, d_out_taxon + r0, d_out_status + r0, ls);
        else rc = aggregate_csr_counted_dev(tx, method, strategy, factor, lower_bound, ranked_only, d_hits, D.total, d_read_pieces, nr, d_out_taxon + r0, d_out_status + r0, ls);
        if (!rc && record_agg) UMGAP_HIP(hipEventRecord(record_agg, ls));
        return rc;
    };
    // UMGAP_TRYP_CHUNKS=C (an experiment that is kept, off by default): the call cut into C chunks of whole reads that alternate
    // between two lanes (streams of the handle, two priorities = two hardware queues), so that one chunk's digest or aggregate could
    // run beside the other's probe.  Measured (round 5, tools/ab_tryp_chunks.py, profiles/r05_tryp_chunks_ab.json: 10 M pairs, 400 M
    // keys): 3.08-3.15 ms unchunked, 3.07 / 3.12 / 3.24 / 3.36 ms with 2 / 4 / 6 / 8 chunks -- nothing to gain: the probe holds four
    // of a SIMD's wave slots and four fifths of its registers, what runs beside it gets one wave per SIMD, and the per-chunk launches
    // and round trip cost what little overlap there is.
    size_t chunks = 1;
    if (const char* e = getenv("UMGAP_TRYP_CHUNKS")) { const long v = atol(e); if (v >= 1 && v <= 64) chunks = (size_t)v; }
    chunks = std::min(chunks, n_reads);
    if (chunks <= 1) return run_chunk(0, n_reads, s, 0, nullptr, nullptr, nullptr);
    if (!ix->tryp_lane[0]) {
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);       // (lo = least, hi = greatest priority; numerically hi <= lo)
        hipStream_t l0 = nullptr, l1 = nullptr;
        hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
        bool ok = hipStreamCreateWithPriority(&l0, hipStreamNonBlocking, hi) == hipSuccess && hipStreamCreateWithPriority(&l1, hipStreamNonBlocking, lo) == hipSuccess;
        for (int i = 0; i < 4 && ok; i++) ok = hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) == hipSuccess;
        if (!ok) {
            if (l0) (void)hipStreamDestroy(l0);
            if (l1) (void)hipStreamDestroy(l1);
            for (auto e : ev) if (e) (void)hipEventDestroy(e);
            return fail(UMGAP_ERR_DEVICE, "umgap_classify_tryp_dev: cannot create the lanes of a chunked call");
        }
        ix->tryp_lane[0] = l0; ix->tryp_lane[1] = l1;
        ix->tryp_ev_agg[0] = ev[0]; ix->tryp_ev_agg[1] = ev[1]; ix->tryp_ev_done[0] = ev[2]; ix->tryp_ev_done[1] = ev[3];
    }
    // the chunk boundaries in bytes, one round trip for all; it also puts the caller's earlier work (the inputs) before the lanes
    std::vector<uint64_t> ends(chunks + 1);
    auto first_read = [&](size_t c) { return (size_t)((unsigned __int128)n_reads * c / chunks); };
    for (size_t c = 0; c <= chunks; c++) UMGAP_HIP(hipMemcpyAsync(&ends[c], d_frag_offsets + first_read(c) * fragments_per_read, 8, hipMemcpyDeviceToHost, s));
    UMGAP_HIP(hipStreamSynchronize(s));
    for (size_t c = 0; c < chunks; c++)
        if (ends[c + 1] < ends[c]) return fail(UMGAP_ERR_INVALID, "tryptic digest: fragment offsets must not decrease");
    rc = UMGAP_OK;
    for (size_t c = 0; c < chunks && !rc; c++) {
        const int l = (int)(c & 1);
        rc = run_chunk(first_read(c), first_read(c + 1), ix->tryp_lane[l], l, &ends[c], c ? ix->tryp_ev_agg[1 - l] : nullptr, ix->tryp_ev_agg[l]);
    }
    if (rc) {       // leave nothing of this call in flight behind an error
        (void)hipStreamSynchronize(ix->tryp_lane[0]);
        (void)hipStreamSynchronize(ix->tryp_lane[1]);
        return rc;
    }
    for (int l = 0; l < 2; l++) {
        UMGAP_HIP(hipEventRecord(ix->tryp_ev_done[l], ix->tryp_lane[l]));
        UMGAP_HIP(hipStreamWaitEvent(s, ix->tryp_ev_done[l], 0));
    }
    return UMGAP_OK;
}

}  // extern "C"
