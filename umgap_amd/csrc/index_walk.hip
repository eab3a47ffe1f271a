// index_walk.hip — the 9-mer index of a proteome laid out in WALK order (kind 10; round 3).
//
// Consecutive 9-mers of a protein fragment share eight residues, yet a hash table sends each of them to a bucket of its own: one
// 64-byte request per window, and ~50 G requests/s is the chip's ceiling.  When the index is built from a
// proteome (umgap_walk_builder_* in index.hip: values from any 9-mer table, order from the proteins; UMGAP's own index is derived
// from UniProt proteins), "the order in which the proteome lists its windows" is a ready-made walk order:
//     E[p]   = taxon of the key | (low 31 key bits | continues << 31) << 32                   8 bytes per entry
//              one entry per key of the index, at the FIRST window of the proteome that holds it, in proteome order (plus short
//              runs of repeated 9-mers inside a protein, which keep entries nothing points at so that the walk is not cut, and
//              the keys no protein shows as walks of one at the end)
//              ("continues": entry p - 1 is the previous window of the same protein, so key(p) = key(p - 1) << 5 | last residue)
//     table  = the exact hash table of common.hpp with value = p of the key's entry
// A read fragment whose windows follow the proteome is then answered by ONE probe (its first window, the anchor, gives p) plus the
// few sectors of E[p + 1 ..]: window j behind the anchor is answered by E[p + j] iff every entry on the way continues its
// predecessor and carries the window's low key bits -- by induction its key then EQUALS the window's, so the answer is exact (no
// fingerprints).  Where the read leaves the walk (a mutation, a window the proteome lists elsewhere first) the next window is
// probed again and walks on from its own position.  The fused surface-B kernels do this (classify.hip); the plain get() calls
// take the position from the table and the taxon from E (two passes).
// Measured first as a prototype for fragments of one fixed length (kept below as umgap_walkproto_lookup: lane = window by plain
// arithmetic, anchors probed lane-per-key): 840 M windows of reads that are proteome fragments with 0 / 1 / 3 / 6 % of their
// residues changed in 7.6 / 8.8 / 12.5 / 16.8 ms, bit-identical to the plain table's answers (which then took 29 ms, most of it a
// binary search per window that window_locate.hpp has since replaced: 17.5 ms).  The product kernels: profiles/r03_walk_index.json.
#include <algorithm>

#include "probe_common.hpp"
#include "walk_ref.hpp"
#include "window_locate.hpp"

using namespace umgap;
using namespace umgap::probe;

namespace {

// key of window w of fragment f (fragments of flen bytes back to back); false: a byte outside 'A'..'Z'
__device__ __forceinline__ bool window_key(const uint8_t* __restrict__ bytes, uint64_t f, uint32_t w, uint32_t flen, uint64_t& key) {
    const uint8_t* p = bytes + f * flen + w;
    key = 0;
    bool ok = true;
#pragma unroll
    for (int c = 0; c < 9; c++) {
        const uint32_t ch = (uint32_t)p[c] - 'A';
        ok = ok && ch < 26u;
        key = (key << 5) | (ch & 31u);
    }
    return ok;
}

// the table's answers for `keys` (vals) -> taxa
__global__ __launch_bounds__(256) void k_walk_map(WalkRef W, const uint64_t* __restrict__ keys, uint32_t* __restrict__ vals, size_t n, uint32_t miss) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += step) vals[i] = walk_taxon(W, vals[i], keys[i], miss);
}

// hits[gw] for every window gw of the reads (fragments of flen residues): a wave takes 64 consecutive windows
__global__ __launch_bounds__(256) void k_walk_lookup(TableView H2, const uint64_t* __restrict__ E64, uint64_t n_E, const uint8_t* __restrict__ reads, uint64_t n_frag,
                                                     uint32_t flen, uint32_t anchor_every, uint32_t* __restrict__ hits, unsigned long long* __restrict__ stats) {
    const uint32_t W = flen - 8, lane = threadIdx.x & 63;
    const uint2* E = reinterpret_cast<const uint2*>(E64);       // x = taxon, y = low key bits | continues << 31
    const uint64_t total = n_frag * W;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    unsigned long long n_walked = 0, n_probed = 0;
    for (uint64_t g0 = wave * 64; g0 < total; g0 += nwaves * 64) {
        const uint64_t gw = g0 + lane;
        const uint64_t f = gw < total ? gw / W : 0;
        const uint32_t w = gw < total ? (uint32_t)(gw - f * W) : 0u;
        uint64_t key = 0;
        const bool valid = gw < total && window_key(reads, f, w, flen, key);
        // distance to the window's anchor: every anchor_every-th window of a fragment, and the wave's first lane (so that no walk
        // starts in another wave)
        const uint32_t j = min(w % anchor_every, lane);
        // 1. anchors: one H2 probe each
        uint32_t pos = UMGAP_NONE;
        if (valid && j == 0) pos = lane_get(H2, key, UMGAP_NONE);
        // 2. the windows behind an anchor: E[anchor position + j]
        const int a = (int)lane - (int)j;                            // the anchor's lane
        const uint32_t apos = (uint32_t)__shfl((int)pos, a, 64);
        bool cand = gw < total && apos != UMGAP_NONE && (uint64_t)apos + j < n_E;
        uint2 e{0u, 0u};
        if (cand) e = E[(uint64_t)apos + j];
        const bool ok_self = cand && (e.y & 0x7FFFFFFFu) == ((uint32_t)key & 0x7FFFFFFFu) && valid && (j == 0 || (e.y >> 31));
        const uint64_t okm = __ballot(ok_self);
        bool walked = false;
        if (cand) {
            const uint64_t need = ((j + 1u >= 64u ? ~0ull : ((1ull << (j + 1)) - 1ull)) << a);   // lanes a .. lane
            walked = (okm & need) == need;
        }
        uint32_t hit = 0;
        if (walked) hit = e.x;
        else if (valid) {
            // 3. not reached by a walk: its own probe, then its entry
            const uint32_t p = j == 0 ? pos : lane_get(H2, key, UMGAP_NONE);
            if (p != UMGAP_NONE && p < n_E) hit = E[p].x;
        }
        if (gw < total) hits[gw] = hit;
        n_walked += (walked && j != 0) ? 1ull : 0ull;
        n_probed += (valid && (j == 0 || !walked)) ? 1ull : 0ull;
    }
    if (stats) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            n_walked += (unsigned long long)(uint32_t)__shfl_xor((int)(uint32_t)n_walked, d, 64);
            n_probed += (unsigned long long)(uint32_t)__shfl_xor((int)(uint32_t)n_probed, d, 64);
        }
        if (lane == 0) { atomicAdd(&stats[0], n_walked); atomicAdd(&stats[1], n_probed); }
    }
}

// The get() of protein windows (umgap_index_lookup_prot9_dev) over a walk index, window-parallel: lane = window g of the flat
// list (fragment found by binary search in the output offsets, as the plain kernel does).  Every 8th window of a fragment, and the
// wave's first lane, is an ANCHOR: one quad-probe batch gives their positions; every other window reads the entry `its distance`
// behind its anchor's position and is answered by it iff all entries from the anchor on continue their predecessor and carry
// their window's low key bits (one ballot).  Windows not reached that way take a second batch of probes and their entry.
__device__ __forceinline__ uint32_t lane_result(const uint32_t (&res)[4], uint32_t lane) {
    // the answer to lane L's key sits, quad-uniform, in quad L & 15 as round L >> 4
    const int src = 4 * (int)(lane & 15u);
    const uint32_t v0 = (uint32_t)__shfl((int)res[0], src, 64), v1 = (uint32_t)__shfl((int)res[1], src, 64),
                   v2 = (uint32_t)__shfl((int)res[2], src, 64), v3 = (uint32_t)__shfl((int)res[3], src, 64);
    const uint32_t r4 = lane >> 4;
    return r4 == 0 ? v0 : (r4 == 1 ? v1 : (r4 == 2 ? v2 : v3));
}

__global__ __launch_bounds__(256) void k_walk_lookup_prot9(TableView H2, WalkRef W, const uint8_t* __restrict__ bytes,
                                                           const uint64_t* __restrict__ offsets, const uint64_t* __restrict__ out_offsets, size_t n_prot,
                                                           size_t n_windows, uint32_t miss, uint32_t* __restrict__ out) {
    constexpr uint32_t EVERY = 8;
    __shared__ uint32_t lds_tmp[4][192];
    const uint32_t lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
    // a wave takes a contiguous range of windows (a multiple of 64): the fragment cursor moves along with it (window_locate.hpp)
    const size_t per_wave = ((n_windows + nwaves - 1) / nwaves + 63) & ~(size_t)63;
    const size_t first = wave * per_wave, last = min(n_windows, first + per_wave);
    if (first >= n_windows) return;
    WindowCursor cur = window_cursor(out_offsets, offsets, n_prot, first);
    for (size_t base = first; base < last; base += 64) {
        const size_t g = base + lane;
        uint64_t key = 0;
        bool valid = false;
        uint32_t w = 0;
        uint64_t ww = 0;
        const uint64_t start = locate_window<true>(cur, out_offsets, offsets, n_prot, base, lane, lds_tmp[threadIdx.x >> 6], ww);   // every lane takes part
        if (g < n_windows) {
            w = ww > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)ww;
            const uint8_t* s = bytes + start;
            valid = true;
#pragma unroll
            for (int c = 0; c < 9; c++) {
                const uint32_t ch = (uint32_t)s[c] - 'A';
                valid = valid && ch < 26u;
                key = (key << 5) | (ch & 31u);
            }
        }
        const uint32_t j = min(w % EVERY, lane);                     // distance to the window's anchor (same fragment, same wave)
        // 1. the anchors
        uint32_t local = 0, tag = 0, ok1 = 0;
        if (valid && j == 0) ok1 = locate(H2, key, local, tag) ? 1u : 0u;
        uint32_t res[4];
        wave_probe64_vals(H2, local, tag | (ok1 << 31), UMGAP_NONE, lane, res);
        const uint64_t* E = W.E;
        const uint64_t n_E = W.n;
        const uint32_t q1 = lane_result(res, lane);                   // (shuffles: every lane takes part)
        const uint64_t pos = (valid && j == 0) ? walk_position(W, q1, key) : WALK_NOPOS;
        // 2. the windows behind them
        const int a = (int)lane - (int)j;
        const uint64_t apos = (uint64_t)(uint32_t)__shfl((int)(uint32_t)pos, a, 64) | ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(pos >> 32), a, 64) << 32);
        const bool cand = g < n_windows && apos != WALK_NOPOS && apos + j < n_E;
        uint64_t e = 0;
        if (cand) e = E[apos + j];
        const uint32_t ey = (uint32_t)(e >> 32);
        const bool ok_self = cand && valid && (ey & 0x7FFFFFFFu) == ((uint32_t)key & 0x7FFFFFFFu) && (j == 0 || (ey >> 31));
        const uint64_t okm = __ballot(ok_self);
        const uint64_t need = (((1ull << (j + 1)) - 1ull) << a);       // lanes a .. lane (j < 8)
        const bool walked = cand && (okm & need) == need;
        // 3. the rest: their own probe (anchors that missed have their answer already), then their entry
        uint32_t local2 = 0, tag2 = 0, ok2 = 0;
        const bool again = valid && !walked && j != 0;
        if (again) ok2 = locate(H2, key, local2, tag2) ? 1u : 0u;
        uint32_t hit = miss;
        if (walked) { hit = (uint32_t)e; if (hit == 0u && W.zero_absent) hit = miss; }    // (a 9-mer of the proteome that is no key)
        if (__any(again)) {                                            // wave-uniform
            wave_probe64_vals(H2, local2, tag2 | (ok2 << 31), UMGAP_NONE, lane, res);
            const uint32_t q2 = lane_result(res, lane);
            if (again) hit = walk_taxon(W, q2, key, miss);
        }
        if (g < n_windows) out[g] = hit;
    }
}

}  // namespace

namespace umgap {
int walk_lookup_prot9(const umgap_index* ix, const uint8_t* d_bytes, const uint64_t* d_offsets, const uint64_t* d_out_offsets, size_t n_prot, size_t n_windows,
                      uint32_t miss, uint32_t* d_out, hipStream_t s) {
    if (n_windows) hipLaunchKernelGGL(k_walk_lookup_prot9, dim3((unsigned)std::max<size_t>(1, std::min<size_t>((n_windows + 255) / 256, 256 * 8))), dim3(256), 0, s, ix->view,
                                      walk_ref_of(ix), d_bytes, d_offsets, d_out_offsets, n_prot, n_windows, miss, d_out);
    UMGAP_HIP(hipGetLastError());
    return UMGAP_OK;
}
// d_vals[i] = the table's answer for d_keys[i] (k_lookup9 with miss = UMGAP_NONE) -> the key's taxon, or `miss`
int walk_map_positions(const umgap_index* ix, const uint64_t* d_keys, uint32_t* d_vals, size_t n, uint32_t miss, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_walk_map, dim3((unsigned)std::max<size_t>(1, std::min<size_t>((n + 255) / 256, 256 * 16))), dim3(256), 0, s, walk_ref_of(ix), d_keys, d_vals, n, miss);
    UMGAP_HIP(hipGetLastError());
    return UMGAP_OK;
}
}  // namespace umgap

extern "C" {

// how a walk-order index numbers its entries (walk_ref.hpp): the table's shift, the entries of the walk array, the exception list
int umgap_index_walk_info(const umgap_index* ix, uint32_t* shift, uint64_t* n_entries, uint32_t* n_exceptions) {
    if (!ix || ix->kind != 10) return fail(UMGAP_ERR_INVALID, "umgap_index_walk_info: not a walk-order index");
    if (shift) *shift = ix->walk_shift;
    if (n_entries) *n_entries = ix->n_walk;
    if (n_exceptions) *n_exceptions = ix->n_walk_exc;
    return UMGAP_OK;
}

// EXPERIMENT kept beside the product (tools/bench_walk_lookup.py; not in the public header): hits of every window of reads made of
// fragments of exactly flen residues, anchors every `anchor_every` windows; d_stats (nullable): {windows answered by a walk, probes made}
int umgap_walkproto_lookup(umgap_index* ix, const uint8_t* d_reads, uint64_t n_frag, uint32_t flen, uint32_t anchor_every,
                           uint32_t* d_hits, unsigned long long* d_stats, void* stream) {
    if (!ix || ix->kind != 10 || !d_reads || !d_hits || flen < 9 || anchor_every < 1 || anchor_every > 32) return fail(UMGAP_ERR_INVALID, "umgap_walkproto_lookup: bad argument");
    if (ix->walk_shift) return fail(UMGAP_ERR_INVALID, "umgap_walkproto_lookup: the prototype reads 32-bit entry numbers (an index of fewer than 2^32 - 2^22 entries)");
    int rc = use_device(ix->device);
    if (rc) return rc;
    hipLaunchKernelGGL(k_walk_lookup, dim3(256 * 8), dim3(256), 0, (hipStream_t)stream, ix->view, ix->d_walk, ix->n_walk, d_reads, n_frag, flen, anchor_every, d_hits, d_stats);
    UMGAP_HIP(hipGetLastError());
    return UMGAP_OK;
}

}  // extern "C"
