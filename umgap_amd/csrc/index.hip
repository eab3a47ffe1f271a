// index.hip — the 9-mer -> taxon table in HBM: build, save/open, and the probe kernels.
// Replaces fst::Map::{from_path,from_bytes,get} (call sites: src/commands/pept2lca.rs:74-79,93;
// src/commands/prot2kmer2lca.rs:109-114,176) and fst::MapBuilder (src/commands/buildindex.rs:38-45).
// Layout: common.hpp.  gfx950 only (wave64, DPP quad permutes).
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <thread>
#include <vector>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_select.hpp>

#include "host_copy.hpp"
#include "probe_common.hpp"
#include "window_locate.hpp"
#include "taxonomy.hpp"
#include "walk_ref.hpp"

namespace umgap {
void var_free(umgap_var_index* v);
// index_walk.hip
int walk_map_positions(const umgap_index* ix, const uint64_t* d_keys, uint32_t* d_vals, size_t n, uint32_t miss, hipStream_t s);
int ensure_stage(umgap_index* ix);
int walk_lookup_prot9(const umgap_index* ix, const uint8_t* d_bytes, const uint64_t* d_offsets, const uint64_t* d_out_offsets, size_t n_prot, size_t n_windows,
                      uint32_t miss, uint32_t* d_out, hipStream_t s);
uint64_t var_table_bytes(const umgap_var_index* v);   // slots + heap
int var_save(const umgap_index* ix, FILE* f);
int var_open(umgap_index* ix, const IndexFileHeader& h, FILE* f, const char* path);
}

using namespace umgap;

namespace {

using namespace umgap::probe;

// Stores the four rounds of a wave's probes: the quad leader of round r owns key 16r+q, i.e. 64
// contiguous bytes per round.
__device__ __forceinline__ void wave_probe64(const TableView& T, uint32_t local, uint32_t tagv, uint32_t miss,
                                             uint32_t* __restrict__ out, size_t base, size_t n, uint32_t lane) {
    uint32_t res[4];
    wave_probe64_vals(T, local, tagv, miss, lane, res);
    // res[r] is quad-uniform: lane j of quad q takes round j, i.e. key 16 j + q -- ONE store instruction for the wave's 256 bytes (it was
    // four of 16 lanes each), and non-temporal: a plain store leaves a dirty line in L2 that is written back later, an `nt` (or sc1) store
    // goes through -- of the 3.5 ms a 4-byte write stream costs beside 840 M random reads, 1.3 ms are that (profiles/r05_gather_stream_variants.json)
    const uint32_t q = lane >> 2, j = lane & 3;
    const uint32_t v = j == 0 ? res[0] : (j == 1 ? res[1] : (j == 2 ? res[2] : res[3]));
    const size_t i = base + 16 * j + q;
    if (i < n) __builtin_nontemporal_store(v, out + i);
}

// ------------------------------------------------------------------------------------------
// lookup kernels
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_lookup9(const uint64_t* __restrict__ keys, size_t n, uint32_t miss,
                                                 uint32_t* __restrict__ out, TableView T) {
    const uint32_t lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (size_t base = wave * 64; base < n; base += nwaves * 64) {
        size_t i = base + lane;
        uint32_t local = 0, tag = 0, valid = 0;
        if (i < n) {
            uint64_t key = __builtin_nontemporal_load(keys + i);
            valid = locate(T, key, local, tag) ? 1u : 0u;
        }
        wave_probe64(T, local, tag | (valid << 31), miss, out, base, n, lane);
    }
}

// Lane-per-key variant (each lane reads its whole 64-byte bucket with four 16-byte loads).
// Kept for the A/B measurement in DESIGN.md; k_lookup9 is the shipped one.
__global__ __launch_bounds__(256) void k_lookup9_lane(const uint64_t* __restrict__ keys, size_t n, uint32_t miss,
                                                      uint32_t* __restrict__ out, TableView T) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        uint32_t local, tag;
        uint32_t res = miss;
        if (locate(T, keys[i], local, tag)) {
            const uint4* p = reinterpret_cast<const uint4*>(T.slots) + (size_t)local * 4;
            uint32_t want = tag;
            for (uint32_t d = 0;; d++) {
                uint4 e0 = p[0], e1 = p[1], e2 = p[2], e3 = p[3];
                bool hit = false;
#define UMGAP_CMP(E)                                                              \
    if ((E.x & TAG_MASK) == want) { res = E.y; hit = true; }                      \
    if ((E.z & TAG_MASK) == want) { res = E.w; hit = true; }
                UMGAP_CMP(e0) UMGAP_CMP(e1) UMGAP_CMP(e2) UMGAP_CMP(e3)
#undef UMGAP_CMP
                bool ovf = e0.x != 0xFFFFFFFFu && (e0.x >> 31);
                if (hit || !ovf) break;
                if (d == MAX_DISP) { if (T.n_stash) res = stash_lookup(T, local + T.home_lo, tag, miss); break; }
                p += 4;
                want += 1u << R_BITS;
            }
        }
        out[i] = res;
    }
}

// prot2kmer2lca surface: thread per output window; the peptide of a window is found by binary
// search in out_offsets, the 9 residues are packed from the byte stream.
__global__ __launch_bounds__(256) void k_lookup_prot9(const uint8_t* __restrict__ bytes,
                                                      const uint64_t* __restrict__ offsets,
                                                      const uint64_t* __restrict__ out_offsets, size_t n_prot,
                                                      size_t n_windows, uint32_t miss, uint32_t* __restrict__ out,
                                                      TableView T) {
    __shared__ uint32_t lds_tmp[4][192];
    const uint32_t lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
    // a wave takes a contiguous range of windows (a multiple of 64): the fragment cursor moves along with it (window_locate.hpp)
    const size_t per_wave = ((n_windows + nwaves - 1) / nwaves + 63) & ~(size_t)63;
    const size_t first = wave * per_wave, last = min(n_windows, first + per_wave);
    if (first >= n_windows) return;
    WindowCursor cur = window_cursor(out_offsets, offsets, n_prot, first);
    for (size_t base = first; base < last; base += 64) {
        size_t g = base + lane;
        uint32_t local = 0, tag = 0, valid = 0;
        uint64_t w = 0;
        const uint64_t start = locate_window<false>(cur, out_offsets, offsets, n_prot, base, lane, lds_tmp[threadIdx.x >> 6], w);   // every lane takes part
        if (g < n_windows) {
            const uint8_t* s = bytes + start;
            uint64_t key = 0;
            bool ok = true;
#pragma unroll
            for (int c = 0; c < 9; c++) {
                uint32_t ch = s[c];
                ok = ok && (ch >= 'A' && ch <= 'Z');
                key = (key << 5) | ((ch - 'A') & 31u);
            }
            valid = (ok && locate(T, key, local, tag)) ? 1u : 0u;
        }
        wave_probe64(T, local, tag | (valid << 31), miss, out, base, n_windows, lane);
    }
}

__global__ __launch_bounds__(256) void k_owner9(const uint64_t* __restrict__ keys, size_t n, uint32_t m, uint32_t nb,
                                                uint32_t n_shards, uint32_t* __restrict__ owner) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        uint32_t b, tag;
        probe_of_key(keys[i], m, b, tag);
        // shard s owns home buckets [nb*s/S, nb*(s+1)/S)
        uint32_t s = (uint32_t)(((uint64_t)b * n_shards) / nb);
        while ((uint64_t)nb * s / n_shards > b) s--;
        while ((uint64_t)nb * (s + 1) / n_shards <= b) s++;
        owner[i] = s;
    }
}

// ------------------------------------------------------------------------------------------
// build kernels
// ------------------------------------------------------------------------------------------
struct BuildFlags { unsigned long long too_full, duplicate, bad_key, n_stash, n_kept_noted; };
constexpr uint32_t STASH_CAP = 1u << 22;  // entries the build may divert before the table is grown

// `slots` points at bucket `lo`.  Whole table: lo = 0, hi = nb, chains wrap to bucket 0 (the halo
// copy is made afterwards).  Shard build: only keys with home in [lo, hi) are inserted and chains
// run on into the shard's own halo (no wrap): each shard is a self-contained valid table.
__device__ __forceinline__ void insert_one(uint64_t* __restrict__ slots, uint32_t m, uint32_t nb, uint32_t lo, uint32_t hi,
                                           uint64_t key, uint32_t val, BuildFlags* flags, uint64_t* __restrict__ stash) {
    if (key >> 45) { atomicAdd(&flags->bad_key, 1ull); return; }
    uint32_t b, tag;
    probe_of_key(key, m, b, tag);
    if (b < lo || b >= hi) return;
    const bool whole = (lo == 0 && hi == nb);
    for (uint32_t d = 0; d <= MAX_DISP; d++) {
        uint32_t bucket = b + d;
        if (whole && bucket >= nb) bucket -= nb;
        uint64_t* s = slots + (size_t)(bucket - lo) * SLOTS;
        uint32_t lo = tag | (d << R_BITS);
        uint64_t entry = ((uint64_t)val << 32) | lo;
        for (uint32_t k = 0; k < SLOTS; k++) {
            unsigned long long old = atomicCAS((unsigned long long*)(s + k), (unsigned long long)EMPTY_SLOT,
                                               (unsigned long long)entry);
            if (old == EMPTY_SLOT) return;
            if (((uint32_t)old & TAG_MASK) == lo) { atomicAdd(&flags->duplicate, 1ull); return; }
        }
        // bucket is full: flag it so probes walk on, then try the next bucket
        atomicOr((unsigned long long*)s, (unsigned long long)OVF_BIT);
    }
    // the whole chain is full: divert to the stash (sorted and de-duplicated on the host afterwards)
    unsigned long long at = atomicAdd(&flags->n_stash, 1ull);
    if (at < STASH_CAP) {
        stash[2 * at] = ((uint64_t)b << R_BITS) | tag;
        stash[2 * at + 1] = val;
    } else {
        atomicAdd(&flags->too_full, 1ull);
    }
}

__global__ __launch_bounds__(256) void k_insert(uint64_t* __restrict__ slots, uint32_t m, uint32_t nb, uint32_t lo, uint32_t hi,
                                                const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vals,
                                                size_t n, BuildFlags* flags, uint64_t* stash) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) insert_one(slots, m, nb, lo, hi, keys[i], vals[i], flags, stash);
}

// synthetic index: key_i = pi(i) written base 20 over ACDEFGHIKLMNPQRSTVWY
__device__ __forceinline__ uint64_t synth_key(uint64_t i, uint64_t a, uint64_t b) {
    const uint64_t M = 512000000000ull;  // 20^9; a < 2^25 and i < 2^39 => no overflow
    uint64_t x = (a * i + b) % M;
    uint64_t key = 0;
#pragma unroll
    for (int j = 0; j < 9; j++) {
        uint32_t d = (uint32_t)(x % 20);
        x /= 20;
        // letter code (c - 'A') of digit d in "ACDEFGHIKLMNPQRSTVWY"
        uint32_t code = d + (d >= 1) + (d >= 8) + (d >= 12) + (d >= 17) + (d >= 19);
        key |= (uint64_t)code << (5 * j);  // least significant digit = last letter = bits [4:0]
    }
    return key;
}

__global__ __launch_bounds__(256) void k_insert_synth(uint64_t* __restrict__ slots, uint32_t m, uint32_t nb, uint32_t lo, uint32_t hi,
                                                      uint64_t n_keys, const uint32_t* __restrict__ taxon_of_row,
                                                      uint64_t n_tax, uint64_t a, uint64_t b, BuildFlags* flags, uint64_t* stash) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < n_keys; i += stride) insert_one(slots, m, nb, lo, hi, synth_key(i, a, b), taxon_of_row[i % n_tax], flags, stash);
}

// Index built from protein fragments: every 9-residue window of fragment i maps to taxa[i]; when the
// same 9-mer occurs in several fragments the SMALLEST taxon wins (deterministic stand-in for the
// LCA that `umgap joinkmers` would compute; only the synthetic proteome uses this).
// windows of every fragment (a fragment of fewer than 9 residues has none) and their exclusive scan: one workgroup walks the list in
// slabs of 4096 counts
__global__ __launch_bounds__(256) void k_window_counts(const uint64_t* __restrict__ off, size_t n_frag, uint32_t* __restrict__ cnt) {
    size_t f = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (; f < n_frag; f += step) {
        const uint64_t len = off[f + 1] - off[f];
        cnt[f] = (off[f + 1] >= off[f] && len >= 9 && len < 0xFFFFFFFFull) ? (uint32_t)(len - 8) : 0u;
    }
}
__global__ __launch_bounds__(1024) void k_scan_counts(const uint32_t* __restrict__ cnt, size_t n, uint64_t* __restrict__ base, uint64_t* __restrict__ total) {
    __shared__ uint32_t wsum[16];
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint64_t run = 0;
    for (size_t s0 = 0; s0 < n; s0 += 4096) {
        const size_t i = s0 + (size_t)threadIdx.x * 4;
        uint32_t v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = i + k < n ? cnt[i + k] : 0u;
        const uint32_t mine = v[0] + v[1] + v[2] + v[3];
        uint32_t incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)incl, o, 64); if (lane >= (uint32_t)o) incl += t; }
        if (lane == 63) wsum[wv] = incl;
        __syncthreads();
        uint32_t before = 0, slab = 0;
        for (uint32_t w = 0; w < 16; w++) { if (w < wv) before += wsum[w]; slab += wsum[w]; }
        uint64_t at = run + before + incl - mine;
#pragma unroll
        for (int k = 0; k < 4; k++) { if (i + k < n) base[i + k] = at; at += v[k]; }
        run += slab;
        __syncthreads();
    }
    if (threadIdx.x == 0) { base[n] = run; *total = run; }
}

__global__ __launch_bounds__(256) void k_insert_windows(uint64_t* __restrict__ slots, uint32_t m, uint32_t nb, const uint8_t* __restrict__ bytes,
                                                        const uint64_t* __restrict__ off, const uint32_t* __restrict__ taxa, size_t n_frag,
                                                        uint32_t max_len, BuildFlags* flags, const uint64_t* __restrict__ wbase) {
    size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x, total = n_frag * max_len;
    for (; g < total; g += stride) {
        size_t i = g / max_len;
        uint32_t w = (uint32_t)(g % max_len);
        uint64_t o = off[i], len = off[i + 1] - o;
        if (len < 9 || w + 9 > len) continue;
        uint64_t key = 0;
        bool ok = true;
        for (int c = 0; c < 9; c++) {
            uint32_t ch = bytes[o + w + c];
            ok = ok && ch >= 'A' && ch <= 'Z';
            key = (key << 5) | ((ch - 'A') & 31u);
        }
        if (!ok) continue;
        uint32_t b, tag, val = taxa ? taxa[i] : (uint32_t)(wbase[i] + w);   // taxa == null: the window's number in the proteome (walk order, index_walk.hip)
        probe_of_key(key, m, b, tag);
        bool done = false;
        for (uint32_t d = 0; d <= MAX_DISP && !done; d++) {
            uint32_t bucket = b + d;
            if (bucket >= nb) bucket -= nb;
            uint64_t* s = slots + (size_t)bucket * SLOTS;
            uint32_t lo = tag | (d << R_BITS);
            uint64_t entry = ((uint64_t)val << 32) | lo;
            for (uint32_t k = 0; k < SLOTS && !done; k++) {
                unsigned long long old = atomicCAS((unsigned long long*)(s + k), (unsigned long long)EMPTY_SLOT, (unsigned long long)entry);
                if (old == EMPTY_SLOT) done = true;
                else if (((uint32_t)old & TAG_MASK) == lo) {
                    // same key already there: keep the smaller taxon (high word only: the low word may gain the overflow flag concurrently)
                    atomicMin(reinterpret_cast<uint32_t*>(s + k) + 1, val);
                    done = true;
                }
            }
            if (!done) atomicOr((unsigned long long*)s, (unsigned long long)OVF_BIT);
        }
        if (!done) atomicAdd(&flags->too_full, 1ull);
    }
}

// ---- the walk-order index (kind 10, index_walk.hip) built from a VALUES table and a proteome handed over in chunks ----
constexpr uint32_t WALK_KEEP_REPEATS = 8;   // a run of up to that many repeated 9-mers inside a fragment keeps its entries (one sector of the walk array)
// T = the position table under construction (whole table, no halo yet: chains wrap; its stash is still a raw list, so a key that
// went there is not seen here -- umgap_walk_builder_finish keeps the first of the entries such a key then gets).
__device__ __forceinline__ uint32_t lane_get_wrap(const uint64_t* __restrict__ slots, uint32_t m, uint32_t nb, uint64_t key, uint32_t miss) {
    uint32_t b, tag;
    probe_of_key(key, m, b, tag);
    for (uint32_t d = 0; d <= MAX_DISP; d++) {
        uint32_t bucket = b + d;
        if (bucket >= nb) bucket -= nb;
        const uint4* p = reinterpret_cast<const uint4*>(slots + (size_t)bucket * SLOTS);
        const uint4 e0 = p[0], e1 = p[1], e2 = p[2], e3 = p[3];
        const uint32_t want = tag | (d << R_BITS);
#define UMGAP_CMP(E)                                   \
    if ((E.x & TAG_MASK) == want) return E.y;          \
    if ((E.z & TAG_MASK) == want) return E.w;
        UMGAP_CMP(e0) UMGAP_CMP(e1) UMGAP_CMP(e2) UMGAP_CMP(e3)
#undef UMGAP_CMP
        if (!(e0.x != 0xFFFFFFFFu && (e0.x >> 31))) return miss;
    }
    return miss;
}
// key of the window at bytes[at .. at + 9); false: a byte outside 'A'..'Z'
__device__ __forceinline__ bool window_key9(const uint8_t* __restrict__ bytes, uint64_t at, uint64_t& key) {
    key = 0;
    bool ok = true;
#pragma unroll
    for (int c = 0; c < 9; c++) {
        const uint32_t ch = (uint32_t)bytes[at + c] - 'A';
        ok = ok && ch < 26u;
        key = (key << 5) | (ch & 31u);
    }
    return ok;
}
// pass 1: the chunk's temporary table C, key -> smallest window number (local to the chunk) among the windows whose key the values
// table holds and the position table does not hold yet: the chunk's candidates for "first window of the proteome with that key"
__global__ __launch_bounds__(256) void k_walk_candidates(uint64_t* __restrict__ slots, uint32_t m, uint32_t nb, const uint8_t* __restrict__ bytes,
                                                         const uint64_t* __restrict__ off, const uint64_t* __restrict__ wbase, size_t n_frag, uint32_t max_len,
                                                         BuildFlags* flags, TableView V, const uint64_t* __restrict__ t_slots, uint32_t t_m, uint32_t t_nb,
                                                         const uint32_t* __restrict__ frag_taxa) {
    size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x, total = n_frag * max_len;
    for (; g < total; g += stride) {
        const size_t i = g / max_len;
        const uint32_t w = (uint32_t)(g % max_len);
        const uint64_t o = off[i], len = off[i + 1] - o;
        if (off[i + 1] < o || len >= 0xFFFFFFFFull || len < 9 || w + 9 > len) continue;   // k_window_counts gave such a fragment no windows
        uint64_t key;
        if (!window_key9(bytes, o + w, key)) continue;
        if (!frag_taxa && lane_get(V, key, UMGAP_NONE) == UMGAP_NONE) continue;        // not a key of the index: answered by a miss of the table
                                                                                       // (frag_taxa: no values table -- every 9-mer of the proteome is a key)
        if (lane_get_wrap(t_slots, t_m, t_nb, key, UMGAP_NONE) != UMGAP_NONE) continue; // an earlier chunk listed it
        const uint32_t val = (uint32_t)(wbase[i] + w);
        uint32_t b, tag;
        probe_of_key(key, m, b, tag);
        bool done = false;
        for (uint32_t d = 0; d <= MAX_DISP && !done; d++) {
            uint32_t bucket = b + d;
            if (bucket >= nb) bucket -= nb;
            uint64_t* s = slots + (size_t)bucket * SLOTS;
            const uint32_t lo = tag | (d << R_BITS);
            const uint64_t entry = ((uint64_t)val << 32) | lo;
            for (uint32_t k = 0; k < SLOTS && !done; k++) {
                unsigned long long old = atomicCAS((unsigned long long*)(s + k), (unsigned long long)EMPTY_SLOT, (unsigned long long)entry);
                if (old == EMPTY_SLOT) done = true;
                else if (((uint32_t)old & TAG_MASK) == lo) { atomicMin(reinterpret_cast<uint32_t*>(s + k) + 1, val); done = true; }
            }
            if (!done) atomicOr((unsigned long long*)s, (unsigned long long)OVF_BIT);
        }
        if (!done) atomicAdd(&flags->too_full, 1ull);
    }
}
// pass 2: flag[local window] = 1 where the window is the one C names for its key (so: a key of the index, new, first in the chunk);
// 3 where it holds a key of the index that is listed elsewhere (a repeat: pass 3 decides whether it keeps an entry of its own)
__global__ __launch_bounds__(256) void k_walk_flags(TableView C, TableView V, const uint8_t* __restrict__ bytes, const uint64_t* __restrict__ off, const uint64_t* __restrict__ wbase,
                                                    size_t n_frag, uint32_t max_len, uint8_t* __restrict__ flag, const uint32_t* __restrict__ frag_taxa) {
    size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x, total = n_frag * max_len;
    for (; g < total; g += stride) {
        const size_t i = g / max_len;
        const uint32_t w = (uint32_t)(g % max_len);
        const uint64_t o = off[i], len = off[i + 1] - o;
        if (off[i + 1] < o || len >= 0xFFFFFFFFull || len < 9 || w + 9 > len) continue;
        const uint32_t local = (uint32_t)(wbase[i] + w);
        uint64_t key;
        uint8_t fl = 0;
        if (window_key9(bytes, o + w, key)) {
            if (lane_get(C, key, UMGAP_NONE) == local) fl = 1;
            else if (frag_taxa || lane_get(V, key, UMGAP_NONE) != UMGAP_NONE) fl = 3;
        }
        flag[local] = fl;
    }
}
// pass 3: a run of at most `gap` repeats between two listed windows of a fragment KEEPS its entries (flag 2: an entry nothing points
// at, there so that a walk through the fragment is not cut by a 9-mer that happens to occur earlier in the proteome); longer runs,
// and runs at the ends of a fragment, are dropped (flag 0: the walk array stays near one entry per key however redundant the
// proteome).  gap = 0 drops every repeat, kept ones of an earlier call included.  cnt[f] = entries of fragment f.
// A fragment's run of entries is padded to a multiple of `group` = 2^shift entries (walk_ref.hpp): cnt[f] counts the padding.
__global__ __launch_bounds__(256) void k_walk_keep_short_repeats(uint8_t* __restrict__ flag, const uint64_t* __restrict__ wbase, size_t n_frag, uint32_t gap, uint32_t* __restrict__ cnt,
                                                                 unsigned long long* __restrict__ n_kept, uint32_t group) {
    size_t f = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (; f < n_frag; f += step) {
        const uint64_t p0 = wbase[f], p1 = wbase[f + 1];
        uint32_t c = 0, kept = 0;
        uint64_t p = p0;
        while (p < p1) {
            const uint8_t fl = flag[p];
            if (fl == 1) { c++; p++; continue; }
            if (fl == 0) { p++; continue; }
            uint64_t q = p;                                   // a run of repeats (3, or 2 from an earlier call) [p, q)
            while (q < p1 && flag[q] >= 2) q++;
            const bool keep = gap && q - p <= gap && p > p0 && flag[p - 1] != 0 && q < p1 && flag[q] == 1;
            for (uint64_t t = p; t < q; t++) flag[t] = keep ? 2 : 0;
            if (keep) { c += (uint32_t)(q - p); kept += (uint32_t)(q - p); }
            p = q;
        }
        cnt[f] = (c + group - 1) & ~(group - 1);
        if (kept) atomicAdd(n_kept, (unsigned long long)kept);
    }
}
// pass 4: a thread walks its fragment; every flagged window gets the next entry of the walk array (taxon from the values table,
// low key bits, "continues" iff the previous window of the fragment was flagged too, i.e. is the entry in front) and, when it is
// the key's first, goes into the position table under that entry's number
// With shift > 0 the table holds at >> shift and the run of a fragment starts at a multiple of 2^shift and is padded to one
// (walk_ref.hpp): the entries of a group come from this thread, which lists every LISTED key that shares its low 31 bits with
// another key of its group in `exc` (key, position pairs; *n_exc counts them, a list longer than exc_cap fails the build).
__global__ __launch_bounds__(256) void k_walk_emit(const uint8_t* __restrict__ bytes, const uint64_t* __restrict__ off, const uint64_t* __restrict__ wbase,
                                                   const uint64_t* __restrict__ fbase, const uint8_t* __restrict__ flag, size_t n_frag, TableView V,
                                                   uint64_t* __restrict__ E, uint64_t e0, uint64_t* __restrict__ t_slots, uint32_t t_m, uint32_t t_nb,
                                                   BuildFlags* flags, uint64_t* __restrict__ stash, uint32_t shift, uint64_t* __restrict__ exc,
                                                   unsigned long long* __restrict__ n_exc, uint64_t exc_cap, const uint32_t* __restrict__ frag_taxa,
                                                   uint64_t* __restrict__ kept_list, uint64_t kept_cap) {
    size_t f = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    const uint64_t gmask = (1ull << shift) - 1ull;
    auto note = [&](uint64_t key, uint64_t pos) {
        const unsigned long long i = atomicAdd(n_exc, 1ull);
        if (i < exc_cap) { exc[2 * i] = key; exc[2 * i + 1] = pos; }
    };
    for (; f < n_frag; f += step) {
        const uint64_t p0 = wbase[f], nw = wbase[f + 1] - p0, o = off[f];
        uint64_t at = e0 + fbase[f];
        bool prev = false;
        uint64_t gkey[3] = {0, 0, 0};      // the entries of the current group written so far (at most 2^shift - 1 = 3)
        uint8_t glisted[3] = {0, 0, 0};
        for (uint64_t w = 0; w < nw; w++) {
            if (!flag[p0 + w]) { prev = false; continue; }
            uint64_t key;
            (void)window_key9(bytes, o + w, key);
            // (frag_taxa: the fragment's own taxon for now; k_walk_min_taxa puts the smallest over all fragments that hold the key)
            const uint32_t taxon = frag_taxa ? frag_taxa[f] : lane_get(V, key, 0u);
            const uint32_t lo = ((uint32_t)key & 0x7FFFFFFFu) | (prev ? 0x80000000u : 0u);
            E[at] = (uint64_t)taxon | ((uint64_t)lo << 32);
            const bool listed = flag[p0 + w] == 1;
            if (!listed && kept_list) {            // a kept repeat: nothing points at it, so its taxon is copied from the key's own entry at the end
                const unsigned long long i = atomicAdd(&flags->n_kept_noted, 1ull);
                if (i < kept_cap) { kept_list[2 * i] = key; kept_list[2 * i + 1] = at; }
            }
            if (listed) insert_one(t_slots, t_m, t_nb, 0u, t_nb, key, (uint32_t)(at >> shift), flags, stash);   // (2: a kept repeat, listed elsewhere)
            if (shift) {
                const uint32_t in_group = (uint32_t)(at & gmask);
                for (uint32_t i = 0; i < in_group; i++) {
                    if (((uint32_t)gkey[i] & 0x7FFFFFFFu) != ((uint32_t)key & 0x7FFFFFFFu) || gkey[i] == key) continue;
                    if (glisted[i]) note(gkey[i], at - in_group + i);
                    if (listed) note(key, at);
                }
                if (in_group < 3) { gkey[in_group] = key; glisted[in_group] = listed ? 1 : 0; }
            }
            at++;
            prev = true;
        }
        while (at & gmask) E[at++] = WALK_PAD_ENTRY;
    }
}
// Builders without a values table, after the index is complete: every window of the proteome lowers the taxon of its key's entry to
// its fragment's (a 32-bit atomic minimum on the entry's low word) -- the value umgap_index_build_windows9_dev keeps on duplicates.
__global__ __launch_bounds__(256) void k_walk_min_taxa(TableView T, WalkRef W, const uint8_t* __restrict__ bytes, const uint64_t* __restrict__ off,
                                                       const uint32_t* __restrict__ frag_taxa, size_t n_frag, uint32_t max_len) {
    size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x, total = n_frag * max_len;
    uint32_t* E32 = reinterpret_cast<uint32_t*>(const_cast<uint64_t*>(W.E));
    for (; g < total; g += stride) {
        const size_t i = g / max_len;
        const uint32_t w = (uint32_t)(g % max_len);
        const uint64_t o = off[i], len = off[i + 1] - o;
        if (off[i + 1] < o || len >= 0xFFFFFFFFull || len < 9 || w + 9 > len) continue;
        uint64_t key;
        if (!window_key9(bytes, o + w, key)) continue;
        const uint32_t q = lane_get(T, key, UMGAP_NONE);
        if (q == UMGAP_NONE) continue;
        const uint64_t p = walk_position(W, q, key);
        if (p != WALK_NOPOS) atomicMin(E32 + 2 * p, frag_taxa[i]);
    }
}
// deferred values: entry of keys[i] <- vals[i]; a key without an entry goes to the leftover list (appended at the end as a walk of one).
// counts: [0] leftovers, [1] entries valued for the first time, [2] keys that came with two different values
__global__ __launch_bounds__(256) void k_walk_set_values(TableView T, WalkRef W, const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vals, size_t n,
                                                         uint64_t* __restrict__ left_keys, uint32_t* __restrict__ left_vals, uint64_t left_cap,
                                                         unsigned long long* __restrict__ counts) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    uint32_t* E32 = reinterpret_cast<uint32_t*>(const_cast<uint64_t*>(W.E));
    for (; i < n; i += step) {
        const uint64_t key = keys[i];
        const uint32_t v = vals[i];
        const uint32_t q = (key >> 45) ? UMGAP_NONE : lane_get(T, key, UMGAP_NONE);
        const uint64_t p = q == UMGAP_NONE ? WALK_NOPOS : walk_position(W, q, key);
        if (p != WALK_NOPOS) {
            const uint32_t old = atomicExch(E32 + 2 * p, v);
            if (old == 0u) atomicAdd(&counts[1], 1ull);
            else if (old != v) atomicAdd(&counts[2], 1ull);
        } else {
            const unsigned long long at = atomicAdd(&counts[0], 1ull);
            if (at < left_cap) { left_keys[at] = key; left_vals[at] = v; }
        }
    }
}
// sorted leftovers: neighbours with one key and two values
__global__ __launch_bounds__(256) void k_walk_left_conflicts(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vals, size_t n, unsigned long long* __restrict__ count) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x + 1;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += step)
        if (keys[i] == keys[i - 1] && vals[i] != vals[i - 1]) atomicAdd(count, 1ull);
}
// ... and the kept repeats (entries nothing points at; list of (key, entry)) take the taxon of their key's own entry
__global__ __launch_bounds__(256) void k_walk_settle_kept(TableView T, WalkRef W, const uint64_t* __restrict__ list, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    uint32_t* E32 = reinterpret_cast<uint32_t*>(const_cast<uint64_t*>(W.E));
    for (; i < n; i += step) {
        const uint64_t key = list[2 * i], at = list[2 * i + 1];
        const uint32_t q = lane_get(T, key, UMGAP_NONE);
        if (q == UMGAP_NONE) continue;
        const uint64_t p = walk_position(W, q, key);
        if (p != WALK_NOPOS) E32[2 * at] = E32[2 * p];
    }
}
__device__ __forceinline__ uint64_t unmix45_dev(uint64_t h, uint64_t inv_c1, uint64_t inv_c2);
// ... and so do the second entries of keys that a full chain sent to the stash twice (list of (stash key form, table value))
__global__ __launch_bounds__(256) void k_walk_settle_dropped(TableView T, WalkRef W, const uint64_t* __restrict__ list, size_t n, uint64_t inv_c1, uint64_t inv_c2) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    uint32_t* E32 = reinterpret_cast<uint32_t*>(const_cast<uint64_t*>(W.E));
    for (; i < n; i += step) {
        const uint64_t sk = list[2 * i];
        const uint64_t key = unmix45_dev((((sk >> R_BITS) / T.m) << R_BITS) | (sk & R_MASK), inv_c1, inv_c2);
        const uint64_t at = walk_position(W, (uint32_t)list[2 * i + 1], key);
        const uint32_t q = lane_get(T, key, UMGAP_NONE);
        if (q == UMGAP_NONE || at == WALK_NOPOS) continue;
        const uint64_t p = walk_position(W, q, key);
        if (p != WALK_NOPOS) E32[2 * at] = E32[2 * p];
    }
}
// inverse of mix45 (common.hpp) with the inverses of its two odd constants handed in
__device__ __forceinline__ uint64_t unmix45_dev(uint64_t h, uint64_t inv_c1, uint64_t inv_c2) {
    h ^= h >> 24;
    h = (h * inv_c2) & MASK45;
    h ^= (h >> 21) ^ (h >> 42);
    h = (h * inv_c1) & MASK45;
    h ^= h >> 23;
    return h;
}
// the keys of the values table (slots, then its stash) that the position table does not hold: the proteome never showed them.
// keys == nullptr: count only.
__global__ __launch_bounds__(256) void k_walk_leftovers(TableView V, TableView T, uint64_t inv_c1, uint64_t inv_c2, uint64_t* __restrict__ keys,
                                                        uint32_t* __restrict__ taxa, unsigned long long* __restrict__ n_out) {
    const size_t n_slots = (size_t)V.nb * SLOTS, total = n_slots + V.n_stash;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (; i < total; i += step) {
        uint64_t h;
        uint32_t taxon;
        if (i < n_slots) {
            const uint64_t e = V.slots[i];
            if (e == EMPTY_SLOT) continue;
            const uint32_t lo = (uint32_t)e, d = (lo >> R_BITS) & 7u, r = lo & R_MASK;
            const uint64_t bucket = i / SLOTS, home = (bucket + V.nb - d) % V.nb;
            h = ((home / V.m) << R_BITS) | r;
            taxon = (uint32_t)(e >> 32);
        } else {
            const uint64_t sk = V.stash_keys[i - n_slots];
            h = (((sk >> R_BITS) / V.m) << R_BITS) | (sk & R_MASK);
            taxon = V.stash_vals[i - n_slots];
        }
        const uint64_t key = unmix45_dev(h, inv_c1, inv_c2);
        if (lane_get(T, key, UMGAP_NONE) != UMGAP_NONE) continue;
        const unsigned long long at = atomicAdd(n_out, 1ull);
        if (keys) { keys[at] = key; taxa[at] = taxon; }
    }
}
// the leftovers as walks of one entry each behind the proteome's
__global__ __launch_bounds__(256) void k_walk_append(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ taxa, size_t n, uint64_t* __restrict__ E, uint64_t e0,
                                                     uint32_t* __restrict__ pos, uint32_t shift, uint64_t* __restrict__ exc, unsigned long long* __restrict__ n_exc,
                                                     uint64_t exc_cap) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += step) {
        const uint64_t at = e0 + i;
        E[at] = (uint64_t)taxa[i] | ((uint64_t)((uint32_t)keys[i] & 0x7FFFFFFFu) << 32);
        pos[i] = (uint32_t)(at >> shift);
        if (shift) {      // (e0 is a multiple of 2^shift: the group's other entries are leftovers too, distinct keys in sorted order)
            const uint64_t g0 = (at >> shift) << shift;
            bool clash = false;
            for (uint64_t q = g0; q < g0 + (1ull << shift); q++)
                if (q != at && q >= e0 && q - e0 < n && ((uint32_t)keys[q - e0] & 0x7FFFFFFFu) == ((uint32_t)keys[i] & 0x7FFFFFFFu)) clash = true;
            if (clash) {
                const unsigned long long k = atomicAdd(n_exc, 1ull);
                if (k < exc_cap) { exc[2 * k] = keys[i]; exc[2 * k + 1] = at; }
            }
        }
    }
}
// first fragment whose windows start at or behind c * per (chunk cuts of a proteome by window count)
__global__ void k_walk_cuts(const uint64_t* __restrict__ wbase, size_t n_frag, uint64_t per, uint32_t n_cuts, uint64_t* __restrict__ cuts) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_cuts) return;
    const uint64_t target = (uint64_t)c * per;
    size_t lo = 0, hi = n_frag;
    while (lo < hi) { const size_t mid = (lo + hi) / 2; if (wbase[mid] < target) lo = mid + 1; else hi = mid; }
    cuts[c] = lo;
}

// Partition a batch of keys by owner shard for the all-to-all exchange: key i goes to slot
// perm[i] = owner*cap + position of `send`; `cursor[o]` counts arrivals per owner.
// A workgroup handles a tile of 4096 keys: positions inside the tile are handed out through LDS
// counters (one ds_add per wave and owner), then ONE global atomic per owner reserves the tile's
// range -- 8 global atomics per 4096 keys instead of one per wave and owner (which serialises on a
// few addresses: 180 ms for 840 M keys).  A tile's keys for one owner land in one compact range,
// so the send writes and the later gather through `perm` stay cache friendly.
constexpr int PART_PER_THREAD = 16;
__global__ __launch_bounds__(256) void k_partition(const uint64_t* __restrict__ keys, size_t n, uint32_t m, uint32_t nb,
                                                   uint32_t n_shards, uint64_t cap, uint64_t* __restrict__ send,
                                                   int32_t* __restrict__ perm, unsigned long long* __restrict__ cursor,
                                                   uint32_t* __restrict__ overflow) {
    __shared__ uint32_t cnt[64];
    __shared__ unsigned long long base[64];
    const uint32_t lane = threadIdx.x & 63;
    const size_t tile = (size_t)blockDim.x * PART_PER_THREAD;
    for (size_t t0 = (size_t)blockIdx.x * tile; t0 < n; t0 += (size_t)gridDim.x * tile) {
        if (threadIdx.x < 64) cnt[threadIdx.x] = 0;
        __syncthreads();
        uint64_t key[PART_PER_THREAD];
        uint32_t own[PART_PER_THREAD], lpos[PART_PER_THREAD];
#pragma unroll
        for (int k = 0; k < PART_PER_THREAD; k++) {
            size_t i = t0 + (size_t)k * blockDim.x + threadIdx.x;
            own[k] = 0xFFFFFFFFu;
            key[k] = 0;
            lpos[k] = 0;
            if (i < n) {
                key[k] = keys[i];
                uint32_t b, tag;
                probe_of_key(key[k], m, b, tag);
                uint32_t s = (uint32_t)(((uint64_t)b * n_shards) / nb);
                while ((uint64_t)nb * s / n_shards > b) s--;
                while ((uint64_t)nb * (s + 1) / n_shards <= b) s++;
                own[k] = s;
            }
            for (uint32_t o = 0; o < n_shards; o++) {     // wave-aggregated LDS reservation
                uint64_t mask = __ballot(own[k] == o);
                if (!mask) continue;
                uint32_t leader = (uint32_t)__builtin_ctzll(mask), wbase = 0;
                if (lane == leader) wbase = atomicAdd(&cnt[o], (uint32_t)__builtin_popcountll(mask));
                wbase = (uint32_t)__shfl((int)wbase, (int)leader, 64);
                if (own[k] == o) lpos[k] = wbase + (uint32_t)__builtin_popcountll(mask & ((1ull << lane) - 1ull));
            }
        }
        __syncthreads();
        if (threadIdx.x < n_shards) base[threadIdx.x] = cnt[threadIdx.x] ? atomicAdd(&cursor[threadIdx.x], (unsigned long long)cnt[threadIdx.x]) : 0ull;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < PART_PER_THREAD; k++) {
            size_t i = t0 + (size_t)k * blockDim.x + threadIdx.x;
            if (own[k] != 0xFFFFFFFFu) {
                uint64_t pos = base[own[k]] + lpos[k];
                if (pos < cap) { send[(size_t)own[k] * cap + pos] = key[k]; perm[i] = (int32_t)((size_t)own[k] * cap + pos); }
                else { perm[i] = 0; atomicExch(overflow, 1u); }
            }
        }
        __syncthreads();
    }
}

// random-sector gather microbenchmark: probe i reads `lanes` x 16 B starting at a bucket chosen
// by a hash of i; xor-folds what it read.
template <int LANES>
__global__ __launch_bounds__(256) void k_gather(const uint64_t* __restrict__ slots, uint32_t nb, size_t n,
                                                uint32_t* __restrict__ out) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = ((size_t)gridDim.x * blockDim.x) / LANES;
    const uint32_t j = threadIdx.x % LANES;
    constexpr uint32_t BUCKETS = (LANES + 3) / 4;
    for (size_t i = t / LANES; i < n; i += stride) {
        uint64_t h = mix45(i * 0x9E3779B97F4A7C15ull >> 19);
        uint32_t bucket = __umulhi((uint32_t)h, nb - BUCKETS + 1);   // multiply-high, not a 64-bit modulo: the benchmark must not be ALU bound
        const u32x4* p = reinterpret_cast<const u32x4*>(slots) + ((size_t)bucket * 4 + j);
        u32x4 e = __builtin_nontemporal_load(p);
        uint32_t x = e.x ^ e.y ^ e.z ^ e.w;
#pragma unroll
        for (int o = 1; o < LANES; o <<= 1) x ^= (uint32_t)__shfl_xor((int)x, o, 64);
        if (j == 0) out[i] = x;
    }
}

// the same with U independent probes in flight per lane (memory-level parallelism like the probe kernels') and no
// store per probe: what was read is xor-folded in a register and written once per lane at the end, so the
// figure is a pure random-read rate (the per-probe stores of k_gather cost it about a fifth)
template <int LANES, int U>
__global__ __launch_bounds__(256) void k_gather_mlp(const uint64_t* __restrict__ slots, uint32_t nb, size_t n,
                                                    uint32_t* __restrict__ out) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = ((size_t)gridDim.x * blockDim.x) / LANES;
    const uint32_t j = threadIdx.x % LANES;
    uint32_t acc = 0;
    for (size_t i = t / LANES; i < n; i += stride * U) {
        u32x4 e[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            size_t ii = i + (size_t)u * stride;
            uint64_t h = mix45(ii * 0x9E3779B97F4A7C15ull >> 19);
            uint32_t bucket = __umulhi((uint32_t)h, nb);
            e[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(slots) + ((size_t)bucket * 4 + j));
        }
#pragma unroll
        for (int u = 0; u < U; u++) acc ^= e[u].x ^ e[u].y ^ e[u].z ^ e[u].w;
    }
    if (t < n) out[t] = acc;
}

// EXPERIMENT (round 5): is there request capacity BESIDE the vector cache?  The ceiling of the gather above is the vector cache's store of
// outstanding misses; scalar loads go through the scalar data cache, a path of its own.  The same gather -- 128
// vector probes per wave step -- plus KS random 64-byte buckets per wave step fetched with s_load_dwordx16 (one bucket = one scalar
// load; the address must be wave-uniform).  If the step takes no longer, the scalar probes came for free.
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
template <int KS>
__global__ __launch_bounds__(256) void k_gather_scalar_mix(const uint64_t* __restrict__ slots, uint32_t nb, size_t n, uint32_t* __restrict__ out) {
    constexpr int LANES = 4, U = 8;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = ((size_t)gridDim.x * blockDim.x) / LANES;
    const uint32_t j = threadIdx.x % LANES;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(t >> 6));
    uint32_t acc = 0, sacc = 0;
    uint32_t step = 0;
    for (size_t i = t / LANES; i < n; i += stride * U, step++) {
        u32x4 e[U];
        u32x16 sv[KS > 0 ? KS : 1];
#pragma unroll
        for (int k = 0; k < KS; k++) {
            const uint64_t hs = mix45((((uint64_t)wave << 24) ^ ((uint64_t)step * KS + k)) * 0xD1B54A32D192ED03ull >> 19);
            const uint32_t b = __umulhi((uint32_t)hs, nb);
            const uint64_t addr = (uint64_t)(uintptr_t)slots + (uint64_t)b * 64;
            const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)addr), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(addr >> 32));
            const uint64_t sa = ((uint64_t)hi << 32) | lo;
            asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=s"(sv[k]) : "s"(sa) : "memory");
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            size_t ii = i + (size_t)u * stride;
            uint64_t h = mix45(ii * 0x9E3779B97F4A7C15ull >> 19);
            uint32_t bucket = __umulhi((uint32_t)h, nb);
            e[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(slots) + ((size_t)bucket * 4 + j));
        }
#pragma unroll
        for (int u = 0; u < U; u++) acc ^= e[u].x ^ e[u].y ^ e[u].z ^ e[u].w;
        if constexpr (KS > 0) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int k = 0; k < KS; k++)
#pragma unroll
                for (int w = 0; w < 16; w++) sacc ^= sv[k][w];
        }
    }
    if (t < n) out[t] = acc ^ sacc;
}
// the same 64-byte quad gather, 8 in flight per lane, with the load's cache-policy bits spelled out (gfx950: sc0, sc1, nt):
// does any policy move the random-sector ceiling?  Loads are inline asm, so the wait is explicit.
template <int POL>
__device__ __forceinline__ u32x4 load_policy(const u32x4* p) {
    u32x4 v;
    if constexpr (POL == 0) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    else if constexpr (POL == 1) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
    else if constexpr (POL == 2) asm volatile("global_load_dwordx4 %0, %1, off sc0" : "=v"(v) : "v"(p) : "memory");
    else if constexpr (POL == 3) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    else if constexpr (POL == 4) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
    else if constexpr (POL == 5) asm volatile("global_load_dwordx4 %0, %1, off sc0 nt" : "=v"(v) : "v"(p) : "memory");
    else if constexpr (POL == 6) asm volatile("global_load_dwordx4 %0, %1, off sc1 nt" : "=v"(v) : "v"(p) : "memory");
    else asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <int POL>
__global__ __launch_bounds__(256) void k_gather_policy(const uint64_t* __restrict__ slots, uint32_t nb, size_t n, uint32_t* __restrict__ out) {
    constexpr int LANES = 4, U = 8;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = ((size_t)gridDim.x * blockDim.x) / LANES;
    const uint32_t j = threadIdx.x % LANES;
    uint32_t acc = 0;
    for (size_t i = t / LANES; i < n; i += stride * U) {
        u32x4 e[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            size_t ii = i + (size_t)u * stride;
            uint64_t h = mix45(ii * 0x9E3779B97F4A7C15ull >> 19);
            uint32_t bucket = __umulhi((uint32_t)h, nb);
            e[u] = load_policy<POL>(reinterpret_cast<const u32x4*>(slots) + ((size_t)bucket * 4 + j));
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < U; u++) acc ^= e[u].x ^ e[u].y ^ e[u].z ^ e[u].w;
    }
    if (t < n) out[t] = acc;
}

// What does a STREAM beside the random reads cost?  The 64-byte quad gather, 8 in flight per lane, plus -- per probe -- SR bytes read
// from and SW bytes written to a side buffer, contiguous per wave step (128 probes: 128 * S bytes as 16-byte lanes), like the key
// and value streams of the lookup and exchange kernels.  (Round 4: the price list behind DESIGN.md's request model.)
template <int SR, int SW>
__global__ __launch_bounds__(256) void k_gather_rw(const uint64_t* __restrict__ slots, uint32_t nb, size_t n, const u32x4* __restrict__ side_in, u32x4* __restrict__ side_out,
                                                   size_t side_vecs, uint32_t* __restrict__ out) {
    constexpr int LANES = 4, U = 8;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = ((size_t)gridDim.x * blockDim.x) / LANES;
    const uint32_t j = threadIdx.x % LANES, lane = threadIdx.x & 63;
    const size_t wave = t >> 6, nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
    uint32_t acc = 0;
    size_t step = 0;
    for (size_t i = t / LANES; i < n; i += stride * U, step++) {
        u32x4 e[U];
        u32x4 r[(SR * 8 + 63) / 64 > 0 ? (SR * 8 + 63) / 64 : 1];
        const size_t chunk = step * nwaves + wave;            // this wave step's 128 probes
        // (the last step of the grid runs its 8 probes per lane past n, so there are up to a step's worth more chunks than n / 128:
        // the streams stop at the end of the side buffers)
        const bool in_r = (chunk + 1) * (SR * 8) <= side_vecs, in_w = (chunk + 1) * (SW * 8) <= side_vecs;
        if constexpr (SR > 0) {
#pragma unroll
            for (int v = 0; v < (SR * 8 + 63) / 64; v++)
                if (in_r && v * 64 + lane < SR * 8u) r[v] = __builtin_nontemporal_load(side_in + chunk * (SR * 8) + v * 64 + lane);
                else r[v] = u32x4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            size_t ii = i + (size_t)u * stride;
            uint64_t h = mix45(ii * 0x9E3779B97F4A7C15ull >> 19);
            uint32_t bucket = __umulhi((uint32_t)h, nb);
            e[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(slots) + ((size_t)bucket * 4 + j));
        }
#pragma unroll
        for (int u = 0; u < U; u++) acc ^= e[u].x ^ e[u].y ^ e[u].z ^ e[u].w;
        if constexpr (SR > 0) {
#pragma unroll
            for (int v = 0; v < (SR * 8 + 63) / 64; v++) acc ^= r[v].x ^ r[v].w;
        }
        if constexpr (SW > 0) {
#pragma unroll
            for (int v = 0; v < (SW * 8 + 63) / 64; v++)
                if (in_w && v * 64 + lane < SW * 8u) side_out[chunk * (SW * 8) + v * 64 + lane] = u32x4{acc, acc + 1u, acc + 2u, acc + 3u};
        }
    }
    if (t < n) out[t] = acc;
}

// The same write stream with the store of step t - 1 issued BEHIND the loads of step t and the wait counted so that it covers the
// loads only (vmcnt(stores)): the wave never waits for a store's acknowledgement before its next loads are evaluated.  If the price
// of a write stream is the in-order return of loads and stores, this gets it back.
template <int SW>
__global__ __launch_bounds__(256) void k_gather_w_deferred(const uint64_t* __restrict__ slots, uint32_t nb, size_t n, u32x4* __restrict__ side_out, size_t side_vecs,
                                                           uint32_t* __restrict__ out) {
    constexpr int LANES = 4, U = 8, NST = (SW * 8 + 63) / 64;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = ((size_t)gridDim.x * blockDim.x) / LANES;
    const uint32_t j = threadIdx.x % LANES, lane = threadIdx.x & 63;
    const size_t wave = t >> 6, nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
    uint32_t acc = 0;
    size_t step = 0;
    bool have_prev = false;
    size_t prev_chunk = 0;
    for (size_t i = t / LANES; i < n; i += stride * U, step++) {
        u32x4 e[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            size_t ii = i + (size_t)u * stride;
            uint64_t h = mix45(ii * 0x9E3779B97F4A7C15ull >> 19);
            uint32_t bucket = __umulhi((uint32_t)h, nb);
            e[u] = load_policy<1>(reinterpret_cast<const u32x4*>(slots) + ((size_t)bucket * 4 + j));
        }
        const u32x4 sv = u32x4{acc, acc + 1u, acc + 2u, acc + 3u};
#pragma unroll
        for (int v = 0; v < NST; v++) {      // the previous step's store (every lane issues NST store instructions, so the count below holds; idle lanes write to a scratch line)
            const bool real = have_prev && (prev_chunk + 1) * (SW * 8) <= side_vecs && v * 64 + lane < SW * 8u;
            u32x4* dst = real ? side_out + prev_chunk * (SW * 8) + v * 64 + lane : side_out + side_vecs + lane;
            asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(dst), "v"(sv) : "memory");
        }
        if constexpr (NST == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
#pragma unroll
        for (int u = 0; u < U; u++) acc ^= e[u].x ^ e[u].y ^ e[u].z ^ e[u].w;
        have_prev = true;
        prev_chunk = step * nwaves + wave;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (t < n) out[t] = acc;
}

// The 4-byte-per-probe write stream as ONE no-return 8-byte atomic exchange per lane and wave step (512 B per 128 probes): an atomic is
// executed at L2 and passes the vector cache as a request of its own kind -- does the write stream's price go with the store path?
__global__ __launch_bounds__(256) void k_gather_w_atomic(const uint64_t* __restrict__ slots, uint32_t nb, size_t n, unsigned long long* __restrict__ side_out, size_t side_words,
                                                         uint32_t* __restrict__ out) {
    constexpr int LANES = 4, U = 8;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = ((size_t)gridDim.x * blockDim.x) / LANES;
    const uint32_t j = threadIdx.x % LANES, lane = threadIdx.x & 63;
    const size_t wave = t >> 6, nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
    uint32_t acc = 0;
    size_t step = 0;
    for (size_t i = t / LANES; i < n; i += stride * U, step++) {
        u32x4 e[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            size_t ii = i + (size_t)u * stride;
            uint64_t h = mix45(ii * 0x9E3779B97F4A7C15ull >> 19);
            uint32_t bucket = __umulhi((uint32_t)h, nb);
            e[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(slots) + ((size_t)bucket * 4 + j));
        }
#pragma unroll
        for (int u = 0; u < U; u++) acc ^= e[u].x ^ e[u].y ^ e[u].z ^ e[u].w;
        const size_t at = (step * nwaves + wave) * 64 + lane;
        if (at < side_words) (void)__hip_atomic_exchange(side_out + at, ((unsigned long long)acc << 32) | acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (t < n) out[t] = acc;
}

// ---- round 5: where does the write stream's price come from?  Three more forms of the 4-byte-per-probe write stream -------------
// (a) MODE 1..4: the same stores with a cache policy on the instruction (sc1 / nt / sc0 sc1 / sc0 sc1 nt: the store-policy table of the
//     MI355X guide); (b) MODE 8: a workgroup's four waves put their values into LDS and ONE wave stores the workgroup's 2 KB as whole
//     lines -- three of four waves never issue a store; (c) MODE 16 + k: a wave keeps 2^k steps' values in registers and stores them
//     together (the same bytes in 2^k times fewer bursts of store instructions).
template <int MODE>
__global__ __launch_bounds__(256) void k_gather_w_variant(const uint64_t* __restrict__ slots, uint32_t nb, size_t n, u32x4* __restrict__ side_out, size_t side_vecs,
                                                          uint32_t* __restrict__ out) {
    constexpr int LANES = 4, U = 8;
    constexpr int AGG = MODE >= 16 ? (1 << (MODE - 16)) : 1;      // steps whose values leave together
    __shared__ u32x4 strip[MODE == 8 ? 128 : 1];                  // MODE 8: 4 waves x 32 lanes x 16 B
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = ((size_t)gridDim.x * blockDim.x) / LANES;
    const uint32_t j = threadIdx.x % LANES, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const size_t wave = t >> 6, nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
    uint32_t acc = 0;
    size_t step = 0;
    u32x4 held[AGG];
    // every thread of the grid runs the SAME number of steps (MODE 8 has a workgroup barrier in the loop): the last step's probes may lie
    // past n -- a probe is a hash of its number, any number will do
    const size_t nsteps = (n + stride * U - 1) / (stride * U);
    for (; step < nsteps; step++) {
        const size_t i = t / LANES + step * stride * U;
        u32x4 e[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            size_t ii = i + (size_t)u * stride;
            uint64_t h = mix45(ii * 0x9E3779B97F4A7C15ull >> 19);
            uint32_t bucket = __umulhi((uint32_t)h, nb);
            e[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(slots) + ((size_t)bucket * 4 + j));
        }
#pragma unroll
        for (int u = 0; u < U; u++) acc ^= e[u].x ^ e[u].y ^ e[u].z ^ e[u].w;
        const u32x4 sv = u32x4{acc, acc + 1u, acc + 2u, acc + 3u};
        const size_t chunk = step * nwaves + wave;                 // 32 vectors (512 B) per wave step
        if constexpr (MODE >= 1 && MODE <= 4) {
            u32x4* dst = side_out + chunk * 32 + lane;
            if (lane < 32u && (chunk + 1) * 32 <= side_vecs) {
                if constexpr (MODE == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(sv) : "memory");
                if constexpr (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(dst), "v"(sv) : "memory");
                if constexpr (MODE == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(sv) : "memory");
                if constexpr (MODE == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(dst), "v"(sv) : "memory");
            }
        } else if constexpr (MODE == 8) {
            if (lane < 32u) strip[wv * 32 + lane] = sv;
            __syncthreads();
            // the workgroup's four chunks are consecutive (waves 4b .. 4b+3): 128 vectors = 2 KB, stored by wave 0 alone, two instructions
            const size_t c0 = step * nwaves + (wave & ~(size_t)3);
            if (wv == 0 && (c0 + 4) * 32 <= side_vecs) {
                side_out[c0 * 32 + lane] = strip[lane];
                side_out[c0 * 32 + 64 + lane] = strip[64 + lane];
            }
            __syncthreads();
        } else {
            held[step % AGG] = sv;
            if ((step % AGG) == AGG - 1) {
                // the wave's AGG chunks are nwaves * 32 vectors apart: AGG stores of 512 B, back to back
#pragma unroll
                for (int k = 0; k < AGG; k++) {
                    const size_t ck = (step - (AGG - 1) + k) * nwaves + wave;
                    if (lane < 32u && (ck + 1) * 32 <= side_vecs) side_out[ck * 32 + lane] = held[k];
                }
            }
        }
    }
    if (t < n) out[t] = acc;
}

__global__ void k_fill(uint64_t* p, size_t n, uint64_t v) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = v;
}

inline uint64_t inv_odd45(uint64_t c) {  // inverse of an odd constant modulo 2^45 (Newton)
    uint64_t x = c;
    for (int i = 0; i < 6; i++) x *= 2 - c * x;
    return x & MASK45;
}
inline int grid_for(size_t n, int per_block = 256, int max_blocks = 256 * 8) {
    size_t g = (n + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > (size_t)max_blocks) g = max_blocks;
    return (int)g;
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
uint32_t choose_m(uint64_t n, double load) {
    if (!(load > 0.0 && load < 1.0)) load = 0.5;
    double buckets = std::ceil((double)n / (SLOTS * load));
    uint64_t m = (uint64_t)std::ceil(buckets / (double)(1u << A_BITS));
    if (m < 1) m = 1;
    return (uint32_t)m;
}

int alloc_table(umgap_index* ix, uint32_t m, uint32_t shard, uint32_t n_shards) {
    uint64_t nb64 = (uint64_t)m << A_BITS;
    if (nb64 + MAX_DISP >= (1ull << 32)) return fail(UMGAP_ERR_INVALID, "index too large for one table (>= 2^32 buckets)");
    uint32_t nb = (uint32_t)nb64;
    uint32_t lo = (uint32_t)((uint64_t)nb * shard / n_shards), hi = (uint32_t)((uint64_t)nb * (shard + 1) / n_shards);
    size_t slots = ((size_t)(hi - lo) + MAX_DISP) * SLOTS;
    if (ix->d_slots) { (void)hipFree(ix->d_slots); ix->d_slots = nullptr; }
    hipError_t e = hipMalloc((void**)&ix->d_slots, slots * sizeof(uint64_t));
    if (e != hipSuccess) return fail(UMGAP_ERR_NOMEM, std::string("hipMalloc of the table failed: ") + hipGetErrorString(e));
    ix->view.slots = ix->d_slots;
    ix->view.m = m;
    ix->view.nb = nb;
    ix->view.first_bucket = lo;
    ix->view.n_resident = hi - lo + MAX_DISP;
    ix->view.home_lo = lo;
    ix->view.home_hi = hi;
    ix->view.stash_keys = nullptr;
    ix->view.stash_vals = nullptr;
    ix->view.n_stash = 0;
    return UMGAP_OK;
}

// install a stash given as host pairs {composite key, taxon}; sorts, refuses duplicates
int set_stash(umgap_index* ix, std::vector<std::pair<uint64_t, uint64_t>>& entries) {
    if (ix->d_stash_keys) { (void)hipFree(ix->d_stash_keys); ix->d_stash_keys = nullptr; }
    if (ix->d_stash_vals) { (void)hipFree(ix->d_stash_vals); ix->d_stash_vals = nullptr; }
    ix->n_stash = 0;
    ix->view.stash_keys = nullptr; ix->view.stash_vals = nullptr; ix->view.n_stash = 0;
    if (entries.empty()) return UMGAP_OK;
    std::sort(entries.begin(), entries.end());
    for (size_t i = 1; i < entries.size(); i++)
        if (entries[i].first == entries[i - 1].first) return fail(UMGAP_ERR_BUILD, "index build: duplicate keys (keys must be distinct, cf. fst::MapBuilder::insert)");
    std::vector<uint64_t> k(entries.size());
    std::vector<uint32_t> v(entries.size());
    for (size_t i = 0; i < entries.size(); i++) { k[i] = entries[i].first; v[i] = (uint32_t)entries[i].second; }
    UMGAP_HIP(hipMalloc((void**)&ix->d_stash_keys, k.size() * 8));
    UMGAP_HIP(hipMalloc((void**)&ix->d_stash_vals, v.size() * 4));
    UMGAP_HIP(hipMemcpy(ix->d_stash_keys, k.data(), k.size() * 8, hipMemcpyHostToDevice));
    UMGAP_HIP(hipMemcpy(ix->d_stash_vals, v.data(), v.size() * 4, hipMemcpyHostToDevice));
    ix->n_stash = (uint32_t)k.size();
    ix->view.stash_keys = ix->d_stash_keys;
    ix->view.stash_vals = ix->d_stash_vals;
    ix->view.n_stash = ix->n_stash;
    return UMGAP_OK;
}

// install the exception list of a walk index (walk_ref.hpp): pairs {key, position}, sorted by key
int set_walk_exceptions(umgap_index* ix, const std::vector<std::pair<uint64_t, uint64_t>>& ex) {
    if (ix->d_walk_exc_keys) { (void)hipFree(ix->d_walk_exc_keys); ix->d_walk_exc_keys = nullptr; }
    if (ix->d_walk_exc_pos) { (void)hipFree(ix->d_walk_exc_pos); ix->d_walk_exc_pos = nullptr; }
    ix->n_walk_exc = 0;
    if (ex.empty()) return UMGAP_OK;
    if (ex.size() >= 0xFFFFFFFFull) return fail(UMGAP_ERR_FORMAT, "walk index: exception list too long");
    std::vector<uint64_t> k(ex.size()), v(ex.size());
    for (size_t i = 0; i < ex.size(); i++) {
        if (i && ex[i].first <= ex[i - 1].first) return fail(UMGAP_ERR_FORMAT, "walk index: exception list not sorted");
        // a listed position is an entry of the walk array (walk_taxon reads E[p]): a truncated or damaged file must not send a probe
        // past the array
        if (ex[i].second >= ix->n_walk) return fail(UMGAP_ERR_FORMAT, "walk index: exception list names entry " + std::to_string(ex[i].second) + " of " + std::to_string(ix->n_walk));
        k[i] = ex[i].first;
        v[i] = ex[i].second;
    }
    if (hipMalloc((void**)&ix->d_walk_exc_keys, k.size() * 8) != hipSuccess || hipMalloc((void**)&ix->d_walk_exc_pos, v.size() * 8) != hipSuccess) {
        (void)hipGetLastError();
        return fail(UMGAP_ERR_NOMEM, "walk index: hipMalloc of the exception list failed");
    }
    UMGAP_HIP(hipMemcpy(ix->d_walk_exc_keys, k.data(), k.size() * 8, hipMemcpyHostToDevice));
    UMGAP_HIP(hipMemcpy(ix->d_walk_exc_pos, v.data(), v.size() * 8, hipMemcpyHostToDevice));
    ix->n_walk_exc = (uint32_t)k.size();
    return UMGAP_OK;
}

// copy buckets [0, MAX_DISP) behind the last bucket so probes never wrap (whole tables only)
int finish_halo(umgap_index* ix) {
    if (!(ix->view.home_lo == 0 && ix->view.home_hi == ix->view.nb)) { UMGAP_HIP(hipStreamSynchronize(ix->stream)); return UMGAP_OK; }
    UMGAP_HIP(hipMemcpyAsync(ix->d_slots + (size_t)ix->view.nb * SLOTS, ix->d_slots, MAX_DISP * SLOTS * sizeof(uint64_t),
                             hipMemcpyDeviceToDevice, ix->stream));
    UMGAP_HIP(hipStreamSynchronize(ix->stream));
    return UMGAP_OK;
}

int new_handle(int device, umgap_index** out) {
    int rc = use_device(device);
    if (rc) return rc;
    umgap_index* ix = new umgap_index();
    ix->device = device;
    hipError_t e = hipStreamCreateWithFlags(&ix->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete ix; return fail(UMGAP_ERR_DEVICE, std::string("hipStreamCreate: ") + hipGetErrorString(e)); }
    // a build reads the caller's device arrays on the handle's own (non-blocking) stream: whatever produced them, on whichever
    // stream, must have finished -- once per handle, before a build that takes seconds
    e = hipDeviceSynchronize();
    if (e != hipSuccess) { (void)hipStreamDestroy(ix->stream); delete ix; return fail(UMGAP_ERR_DEVICE, std::string("hipDeviceSynchronize: ") + hipGetErrorString(e)); }
    *out = ix;
    return UMGAP_OK;
}

// Generic build driver: `insert(ix, flags)` enqueues the insert kernels on ix->stream.
template <class F>
int build_with(umgap_index* ix, uint64_t n, double load, uint32_t shard, uint32_t n_shards, F insert) {
    uint32_t m = choose_m(n, load);
    BuildFlags* d_flags = nullptr;
    uint64_t* d_stash = nullptr;
    UMGAP_HIP(hipMalloc((void**)&d_flags, sizeof(BuildFlags)));
    UMGAP_HIP(hipMalloc((void**)&d_stash, (size_t)STASH_CAP * 16));
    int rc = UMGAP_OK;
    for (int attempt = 0; attempt < 4; attempt++) {
        rc = alloc_table(ix, m, shard, n_shards);
        if (rc) break;
        size_t nslots = (size_t)ix->view.n_resident * SLOTS;
        hipLaunchKernelGGL(k_fill, dim3(grid_for(nslots)), dim3(256), 0, ix->stream, ix->d_slots, nslots, EMPTY_SLOT);
        hipError_t e = hipMemsetAsync(d_flags, 0, sizeof(BuildFlags), ix->stream);
        if (e != hipSuccess) { rc = fail(UMGAP_ERR_DEVICE, hipGetErrorString(e)); break; }
        rc = insert(ix, d_flags, d_stash);
        if (rc) break;
        BuildFlags h{};
        e = hipMemcpyAsync(&h, d_flags, sizeof(h), hipMemcpyDeviceToHost, ix->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ix->stream);
        if (e != hipSuccess) { rc = fail(UMGAP_ERR_DEVICE, std::string("index build: ") + hipGetErrorString(e)); break; }
        if (h.bad_key) { rc = fail(UMGAP_ERR_BUILD, "index build: " + std::to_string(h.bad_key) + " keys are not 45-bit packed 9-mers"); break; }
        if (h.duplicate) { rc = fail(UMGAP_ERR_BUILD, "index build: " + std::to_string(h.duplicate) + " duplicate keys (keys must be distinct, cf. fst::MapBuilder::insert)"); break; }
        if (!h.too_full) {
            std::vector<uint64_t> raw(2 * (size_t)h.n_stash);
            if (h.n_stash && hipMemcpy(raw.data(), d_stash, raw.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(UMGAP_ERR_DEVICE, "index build: stash copy failed"); break; }
            std::vector<std::pair<uint64_t, uint64_t>> entries(h.n_stash);
            for (size_t i = 0; i < h.n_stash; i++) entries[i] = {raw[2 * i], raw[2 * i + 1]};
            rc = set_stash(ix, entries);
            if (!rc) rc = finish_halo(ix);
            break;
        }
        m = m + m / 4 + 1;  // more than STASH_CAP keys did not fit: grow and retry
        rc = fail(UMGAP_ERR_BUILD, "index build: table too full");
    }
    (void)hipFree(d_flags);
    (void)hipFree(d_stash);
    ix->n_keys = n;
    return rc;
}

}  // namespace
// three pinned + three device staging buffers of 24 MB each and the upload stream of the host-pointer look-ups (also index_var.hip)
int umgap::ensure_stage(umgap_index* ix) {
    if (ix->stage_cap) return UMGAP_OK;
    const size_t cap = 2u << 20;  // keys per chunk: 16 MiB keys + 8 MiB results (three slots: 72 MiB pinned, allocated at the first host-pointer call)
    for (int i = 0; i < 3; i++) {
        UMGAP_HIP(hipHostMalloc(&ix->h_stage[i], cap * 12, hipHostMallocDefault));
        UMGAP_HIP(hipMalloc(&ix->d_stage[i], cap * 12));
    }
    if (!ix->up_stream) {
        UMGAP_HIP(hipStreamCreateWithFlags(&ix->up_stream, hipStreamNonBlocking));
        UMGAP_HIP(hipStreamCreateWithFlags(&ix->run_stream, hipStreamNonBlocking));
    }
    ix->stage_cap = cap;
    return UMGAP_OK;
}
namespace {

}  // namespace

// ==========================================================================================
// C ABI
// ==========================================================================================
extern "C" {

int umgap_pack9(const char* s, uint64_t* key45) {
    uint64_t k = 0;
    for (int i = 0; i < 9; i++) {
        unsigned char c = (unsigned char)s[i];
        if (c < 'A' || c > 'Z') return 0;
        k = (k << 5) | (uint64_t)(c - 'A');
    }
    *key45 = k;
    return 1;
}
void umgap_unpack9(uint64_t key45, char* s) {
    for (int i = 0; i < 9; i++) s[i] = (char)('A' + ((key45 >> (5 * (8 - i))) & 31));
}

int umgap_index_build9_dev(const uint64_t* d_keys, const uint32_t* d_taxa, size_t n, double load, int device,
                           umgap_index** out) {
    if (!out || (n && (!d_keys || !d_taxa))) return fail(UMGAP_ERR_INVALID, "umgap_index_build9_dev: null argument");
    umgap_index* ix = nullptr;
    int rc = new_handle(device, &ix);
    if (rc) return rc;
    rc = build_with(ix, n, load, 0, 1, [&](umgap_index* x, BuildFlags* f, uint64_t* st) {
        if (n) hipLaunchKernelGGL(k_insert, dim3(grid_for(n)), dim3(256), 0, x->stream, x->d_slots, x->view.m, x->view.nb, x->view.home_lo, x->view.home_hi, d_keys, d_taxa, n, f, st);
        return UMGAP_OK;
    });
    if (rc) { umgap_index_free(ix); return rc; }
    *out = ix;
    return UMGAP_OK;
}

int umgap_index_build9(const uint64_t* keys, const uint32_t* taxa, size_t n, double load, int device, umgap_index** out) {
    if (!out || (n && (!keys || !taxa))) return fail(UMGAP_ERR_INVALID, "umgap_index_build9: null argument");
    int rc = use_device(device);
    if (rc) return rc;
    uint64_t* dk = nullptr;
    uint32_t* dv = nullptr;
    if (n) {
        UMGAP_HIP(hipMalloc((void**)&dk, n * sizeof(uint64_t)));
        hipError_t e = hipMalloc((void**)&dv, n * sizeof(uint32_t));
        if (e != hipSuccess) { (void)hipFree(dk); return fail(UMGAP_ERR_NOMEM, hipGetErrorString(e)); }
        e = hipMemcpy(dk, keys, n * sizeof(uint64_t), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(dv, taxa, n * sizeof(uint32_t), hipMemcpyHostToDevice);
        if (e != hipSuccess) { (void)hipFree(dk); (void)hipFree(dv); return fail(UMGAP_ERR_DEVICE, hipGetErrorString(e)); }
    }
    rc = umgap_index_build9_dev(dk, dv, n, load, device, out);
    if (dk) (void)hipFree(dk);
    if (dv) (void)hipFree(dv);
    return rc;
}

// ---- incremental build from host chunks (umgap buildindex streams a TSV of any size through this) ----
}  // extern "C"
struct umgap_index_builder {
    umgap_index* ix = nullptr;
    BuildFlags* d_flags = nullptr;
    uint64_t* d_stash = nullptr;
    uint64_t n_added = 0;
    void* h_stage[2] = {nullptr, nullptr};
    void* d_stage[2] = {nullptr, nullptr};
    hipEvent_t done[2] = {nullptr, nullptr};
    bool busy[2] = {false, false};
    size_t cap = 0;
    int slot = 0;
};
extern "C" {

void umgap_index_builder_free(umgap_index_builder* b) {
    if (!b) return;
    if (b->ix) { (void)use_device(b->ix->device); (void)hipStreamSynchronize(b->ix->stream); }
    for (int i = 0; i < 2; i++) {
        if (b->h_stage[i]) (void)hipHostFree(b->h_stage[i]);
        if (b->d_stage[i]) (void)hipFree(b->d_stage[i]);
        if (b->done[i]) (void)hipEventDestroy(b->done[i]);
    }
    if (b->d_flags) (void)hipFree(b->d_flags);
    if (b->d_stash) (void)hipFree(b->d_stash);
    if (b->ix) umgap_index_free(b->ix);
    delete b;
}

int umgap_index_builder_new(uint64_t expected_keys, double load, int device, umgap_index_builder** out) {
    if (!out) return fail(UMGAP_ERR_INVALID, "umgap_index_builder_new: null argument");
    umgap_index_builder* b = new umgap_index_builder();
    int rc = new_handle(device, &b->ix);
    if (rc) { delete b; return rc; }
    auto bail = [&](int code) { umgap_index_builder_free(b); return code; };
    rc = alloc_table(b->ix, choose_m(expected_keys, load), 0, 1);
    if (rc) return bail(rc);
    if (hipMalloc((void**)&b->d_flags, sizeof(BuildFlags)) != hipSuccess || hipMalloc((void**)&b->d_stash, (size_t)STASH_CAP * 16) != hipSuccess)
        return bail(fail(UMGAP_ERR_NOMEM, "umgap_index_builder_new: hipMalloc failed"));
    size_t nslots = (size_t)b->ix->view.n_resident * SLOTS;
    hipLaunchKernelGGL(k_fill, dim3(grid_for(nslots)), dim3(256), 0, b->ix->stream, b->ix->d_slots, nslots, EMPTY_SLOT);
    if (hipMemsetAsync(b->d_flags, 0, sizeof(BuildFlags), b->ix->stream) != hipSuccess) return bail(fail(UMGAP_ERR_DEVICE, "umgap_index_builder_new: memset failed"));
    b->cap = 4u << 20;   // keys per staging slot: 32 MiB of keys + 16 MiB of taxa
    for (int i = 0; i < 2; i++) {
        if (hipHostMalloc(&b->h_stage[i], b->cap * 12, hipHostMallocDefault) != hipSuccess || hipMalloc(&b->d_stage[i], b->cap * 12) != hipSuccess ||
            hipEventCreateWithFlags(&b->done[i], hipEventDisableTiming) != hipSuccess)
            return bail(fail(UMGAP_ERR_NOMEM, "umgap_index_builder_new: staging allocation failed"));
    }
    *out = b;
    return UMGAP_OK;
}

int umgap_index_builder_add(umgap_index_builder* b, const uint64_t* keys45, const uint32_t* taxa, size_t n) {
    if (!b || !b->ix || (n && (!keys45 || !taxa))) return fail(UMGAP_ERR_INVALID, "umgap_index_builder_add: null argument");
    int rc = use_device(b->ix->device);
    if (rc) return rc;
    umgap_index* x = b->ix;
    for (size_t off = 0; off < n; off += b->cap, b->slot ^= 1) {
        const int s = b->slot;
        const size_t len = std::min(b->cap, n - off);
        if (b->busy[s]) { UMGAP_HIP(hipEventSynchronize(b->done[s])); b->busy[s] = false; }
        uint64_t* hk = (uint64_t*)b->h_stage[s];
        uint32_t* hv = (uint32_t*)((char*)b->h_stage[s] + b->cap * 8);
        memcpy(hk, keys45 + off, len * 8);
        memcpy(hv, taxa + off, len * 4);
        uint64_t* dk = (uint64_t*)b->d_stage[s];
        uint32_t* dv = (uint32_t*)((char*)b->d_stage[s] + b->cap * 8);
        UMGAP_HIP(hipMemcpyAsync(dk, hk, len * 8, hipMemcpyHostToDevice, x->stream));
        UMGAP_HIP(hipMemcpyAsync(dv, hv, len * 4, hipMemcpyHostToDevice, x->stream));
        hipLaunchKernelGGL(k_insert, dim3(grid_for(len)), dim3(256), 0, x->stream, x->d_slots, x->view.m, x->view.nb, x->view.home_lo, x->view.home_hi,
                           (const uint64_t*)dk, (const uint32_t*)dv, len, b->d_flags, b->d_stash);
        UMGAP_HIP(hipEventRecord(b->done[s], x->stream));
        b->busy[s] = true;
    }
    b->n_added += n;
    return UMGAP_OK;
}

int umgap_index_builder_finish(umgap_index_builder* b, umgap_index** out) {
    if (!b || !b->ix || !out) return fail(UMGAP_ERR_INVALID, "umgap_index_builder_finish: null argument");
    int rc = use_device(b->ix->device);
    if (rc) return rc;
    umgap_index* x = b->ix;
    BuildFlags h{};
    hipError_t e = hipMemcpyAsync(&h, b->d_flags, sizeof(h), hipMemcpyDeviceToHost, x->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(x->stream);
    if (e != hipSuccess) return fail(UMGAP_ERR_DEVICE, std::string("index build: ") + hipGetErrorString(e));
    if (h.bad_key) return fail(UMGAP_ERR_BUILD, "index build: " + std::to_string(h.bad_key) + " keys are not 45-bit packed 9-mers");
    if (h.duplicate) return fail(UMGAP_ERR_BUILD, "index build: " + std::to_string(h.duplicate) + " duplicate keys (keys must be distinct, cf. fst::MapBuilder::insert)");
    if (h.too_full) return fail(UMGAP_ERR_BUILD, "index build: table too full (more keys than announced?)");
    std::vector<uint64_t> raw(2 * (size_t)h.n_stash);
    if (h.n_stash && hipMemcpy(raw.data(), b->d_stash, raw.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return fail(UMGAP_ERR_DEVICE, "index build: stash copy failed");
    std::vector<std::pair<uint64_t, uint64_t>> entries(h.n_stash);
    for (size_t i = 0; i < h.n_stash; i++) entries[i] = {raw[2 * i], raw[2 * i + 1]};
    rc = set_stash(x, entries);
    if (!rc) rc = finish_halo(x);
    if (rc) return rc;
    x->n_keys = b->n_added;
    *out = x;
    b->ix = nullptr;
    umgap_index_builder_free(b);
    return UMGAP_OK;
}

int umgap_synth_index_build(uint64_t n_keys, const uint32_t* d_taxon_of_row, uint64_t n_tax, uint64_t a, uint64_t b,
                            double load, int device, umgap_index** out) {
    return umgap_synth_index_build_shard(n_keys, d_taxon_of_row, n_tax, a, b, load, device, 0, 1, out);
}

int umgap_synth_index_build_shard(uint64_t n_keys, const uint32_t* d_taxon_of_row, uint64_t n_tax, uint64_t a, uint64_t b,
                                  double load, int device, uint32_t shard, uint32_t n_shards, umgap_index** out) {
    if (!out || !d_taxon_of_row || !n_tax) return fail(UMGAP_ERR_INVALID, "umgap_synth_index_build: null argument");
    if (n_shards == 0 || shard >= n_shards) return fail(UMGAP_ERR_INVALID, "umgap_synth_index_build: shard >= n_shards");
    if (a >= (1ull << 25) || (a % 2) == 0 || (a % 5) == 0 || n_keys > 512000000000ull)
        return fail(UMGAP_ERR_INVALID, "umgap_synth_index_build: need odd a < 2^25 with a % 5 != 0 and n_keys <= 20^9");
    umgap_index* ix = nullptr;
    int rc = new_handle(device, &ix);
    if (rc) return rc;
    rc = build_with(ix, n_keys, load, shard, n_shards, [&](umgap_index* x, BuildFlags* f, uint64_t* st) {
        hipLaunchKernelGGL(k_insert_synth, dim3(grid_for(n_keys, 256, 256 * 16)), dim3(256), 0, x->stream, x->d_slots, x->view.m,
                           x->view.nb, x->view.home_lo, x->view.home_hi, n_keys, d_taxon_of_row, n_tax, a, b % 512000000000ull, f, st);
        return UMGAP_OK;
    });
    if (rc) { umgap_index_free(ix); return rc; }
    *out = ix;
    return UMGAP_OK;
}

void umgap_index_free(umgap_index* ix) {
    if (!ix) return;
    (void)hipSetDevice(ix->device);
    if (ix->bound_to) {   // leave the taxonomy's registry (bind.hip); the values die with the table
        auto& v = ((umgap_taxonomy*)ix->bound_to)->bound_indexes;
        for (size_t i = 0; i < v.size(); i++)
            if (v[i] == ix) { v.erase(v.begin() + (long)i); break; }
    }
    if (ix->var) var_free(ix->var);
    if (ix->d_walk) (void)hipFree(ix->d_walk);
    if (ix->d_walk_exc_keys) (void)hipFree(ix->d_walk_exc_keys);
    if (ix->d_walk_exc_pos) (void)hipFree(ix->d_walk_exc_pos);
    for (void* p : ix->tryp_scratch) if (p) (void)hipFree(p);
    if (ix->tryp_host) (void)hipHostFree(ix->tryp_host);
    for (int l = 0; l < 2; l++) {
        if (ix->tryp_lane[l]) { (void)hipStreamSynchronize(ix->tryp_lane[l]); (void)hipStreamDestroy(ix->tryp_lane[l]); }
        if (ix->tryp_ev_agg[l]) (void)hipEventDestroy(ix->tryp_ev_agg[l]);
        if (ix->tryp_ev_done[l]) (void)hipEventDestroy(ix->tryp_ev_done[l]);
    }
    if (ix->d_slots) (void)hipFree(ix->d_slots);
    if (ix->d_stash_keys) (void)hipFree(ix->d_stash_keys);
    if (ix->d_stash_vals) (void)hipFree(ix->d_stash_vals);
    for (int i = 0; i < 3; i++) {
        if (ix->h_stage[i]) (void)hipHostFree(ix->h_stage[i]);
        if (ix->d_stage[i]) (void)hipFree(ix->d_stage[i]);
    }
    if (ix->up_stream) (void)hipStreamDestroy(ix->up_stream);
    if (ix->run_stream) (void)hipStreamDestroy(ix->run_stream);
    if (ix->stream) (void)hipStreamDestroy(ix->stream);
    delete ix;
}

uint64_t umgap_index_num_keys(const umgap_index* ix) { return ix ? ix->n_keys : 0; }
uint64_t umgap_index_num_buckets(const umgap_index* ix) { return ix ? ix->view.nb : 0; }
uint64_t umgap_index_table_bytes(const umgap_index* ix) {
    if (!ix) return 0;
    return ix->kind == 2 ? var_table_bytes(ix->var) : (uint64_t)ix->view.n_resident * 64;
}
int umgap_index_device(const umgap_index* ix) { return ix ? ix->device : -1; }

int umgap_index_save(const umgap_index* ix, const char* path) {
    if (!ix || !path) return fail(UMGAP_ERR_INVALID, "umgap_index_save: null argument");
    if (ix->bound_to) return fail(UMGAP_ERR_INVALID, "the index is bound to a taxonomy (umgap_index_bind_taxonomy): its values are interval codes, not taxon ids; unbind it first");
    if (ix->kind == 2) {
        int rcv = use_device(ix->device);
        if (rcv) return rcv;
        FILE* fv = fopen(path, "wb");
        if (!fv) return fail(UMGAP_ERR_IO, std::string("cannot create ") + path + ": " + strerror(errno));
        rcv = var_save(ix, fv);
        if (fclose(fv) != 0 && !rcv) rcv = fail(UMGAP_ERR_IO, std::string("error writing ") + path);
        return rcv;
    }
    if (ix->view.home_lo != 0 || ix->view.home_hi != ix->view.nb) return fail(UMGAP_ERR_INVALID, "umgap_index_save: handle holds a shard");
    int rc = use_device(ix->device);
    if (rc) return rc;
    FILE* f = fopen(path, "wb");
    if (!f) return fail(UMGAP_ERR_IO, std::string("cannot create ") + path + ": " + strerror(errno));
    IndexFileHeader h{};
    memcpy(h.magic, "UMGAPIX1", 8);
    h.version = 1; h.kind = (uint32_t)ix->kind; h.n_keys = ix->n_keys; h.m = ix->view.m; h.nb = ix->view.nb;
    // kind 10 (walk order): the walk array, n_walk x u64, follows the stash; bits 60-61 of `reserved` = walk_shift (walk_ref.hpp), and
    // with a shift the exception list follows the walk array: u64 count, the sorted keys, their positions
    h.reserved = ix->kind == 10 ? (ix->n_walk | ((uint64_t)ix->walk_shift << 60) | ((uint64_t)(ix->walk_zero_absent ? 1 : 0) << 62)) : 0;   // bit 62: an entry with value 0 is no key
    h.mix_c1 = MIX_C1; h.mix_c2 = MIX_C2; h.r_bits = R_BITS; h.slots = SLOTS;
    h.n_stash = ix->n_stash;
    bool ok = fwrite(&h, sizeof(h), 1, f) == 1;
    const size_t chunk = 64u << 20;
    std::vector<char> buf(chunk);
    size_t total = (size_t)ix->view.nb * 64;
    for (size_t off = 0; ok && off < total; off += chunk) {
        size_t len = std::min(chunk, total - off);
        if (hipMemcpy(buf.data(), (const char*)ix->d_slots + off, len, hipMemcpyDeviceToHost) != hipSuccess) { ok = false; break; }
        ok = fwrite(buf.data(), 1, len, f) == len;
    }
    if (ok && ix->n_stash) {
        std::vector<uint64_t> k(ix->n_stash), kv(2 * (size_t)ix->n_stash);
        std::vector<uint32_t> v(ix->n_stash);
        ok = hipMemcpy(k.data(), ix->d_stash_keys, k.size() * 8, hipMemcpyDeviceToHost) == hipSuccess &&
             hipMemcpy(v.data(), ix->d_stash_vals, v.size() * 4, hipMemcpyDeviceToHost) == hipSuccess;
        for (size_t i = 0; i < k.size(); i++) { kv[2 * i] = k[i]; kv[2 * i + 1] = v[i]; }
        ok = ok && fwrite(kv.data(), 8, kv.size(), f) == kv.size();
    }
    if (ix->kind == 10) {
        total = (size_t)ix->n_walk * 8;
        for (size_t off = 0; ok && off < total; off += chunk) {
            size_t len = std::min(chunk, total - off);
            if (hipMemcpy(buf.data(), (const char*)ix->d_walk + off, len, hipMemcpyDeviceToHost) != hipSuccess) { ok = false; break; }
            ok = fwrite(buf.data(), 1, len, f) == len;
        }
        if (ok && ix->walk_shift) {
            const uint64_t n_exc = ix->n_walk_exc;
            std::vector<uint64_t> k(n_exc), v(n_exc);
            ok = fwrite(&n_exc, 8, 1, f) == 1;
            if (ok && n_exc)
                ok = hipMemcpy(k.data(), ix->d_walk_exc_keys, n_exc * 8, hipMemcpyDeviceToHost) == hipSuccess &&
                     hipMemcpy(v.data(), ix->d_walk_exc_pos, n_exc * 8, hipMemcpyDeviceToHost) == hipSuccess && fwrite(k.data(), 8, n_exc, f) == n_exc &&
                     fwrite(v.data(), 8, n_exc, f) == n_exc;
        }
    }
    ok = (fclose(f) == 0) && ok;
    if (!ok) return fail(UMGAP_ERR_IO, std::string("error writing ") + path);
    return UMGAP_OK;
}

}  // extern "C"

namespace umgap {
// bytes [file_off, file_off + total) of `fd` -> device memory.  The ~100 GB body of a table is pulled in by several threads, each
// with its own pinned buffer and stream: pread (page cache or NVMe queue depth) of one chunk overlaps the uploads of the others
// (a single read-then-copy loop tops out near 4 GB/s; this runs at 45 GB/s from a warm page cache).  Pinned staging per loader
// thread: 32 MiB for the big tables, less for small ones (pinning memory is not free and the command-line tools open an index
// per invocation).  UMGAP_LOAD_THREADS overrides the 8 loader threads.
bool load_file_range(int fd, off_t file_off, size_t total, void* d_dst, int device) {
    if (!total) return true;
    const size_t chunk = std::min<size_t>(32u << 20, std::max<size_t>(1u << 20, (total / 32 + 4095) & ~(size_t)4095));
    const size_t n_chunks = (total + chunk - 1) / chunk;
    std::atomic<bool> ok{true};
    std::atomic<size_t> next{0};
    auto work = [&]() {
        if (hipSetDevice(device) != hipSuccess) { ok = false; return; }
        void* pinned = nullptr;
        hipStream_t st = nullptr;
        if (hipHostMalloc(&pinned, chunk, hipHostMallocDefault) != hipSuccess) { ok = false; return; }
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { (void)hipHostFree(pinned); ok = false; return; }
        for (size_t c; ok && (c = next.fetch_add(1)) < n_chunks;) {
            const size_t off = c * chunk, len = std::min(chunk, total - off);
            size_t got = 0;
            while (got < len) {
                ssize_t r = pread(fd, (char*)pinned + got, len - got, file_off + (off_t)(off + got));
                if (r <= 0) { ok = false; break; }
                got += (size_t)r;
            }
            if (!ok) break;
            if (hipMemcpyAsync((char*)d_dst + off, pinned, len, hipMemcpyHostToDevice, st) != hipSuccess ||
                hipStreamSynchronize(st) != hipSuccess) { ok = false; break; }
        }
        (void)hipStreamDestroy(st);
        (void)hipHostFree(pinned);
    };
    unsigned n_threads = std::min(8u, std::max(1u, std::thread::hardware_concurrency()));
    if (const char* env = getenv("UMGAP_LOAD_THREADS")) { int v = atoi(env); if (v >= 1) n_threads = (unsigned)std::min(v, 32); }
    n_threads = (unsigned)std::max<size_t>(1, std::min<size_t>(n_threads, n_chunks));
    std::vector<std::thread> th;
    for (unsigned t = 1; t < n_threads; t++) th.emplace_back(work);
    work();
    for (auto& t : th) t.join();
    return ok;
}
}  // namespace umgap

extern "C" {

int umgap_index_open_shard(const char* path, uint32_t device_mask, uint32_t shard, uint32_t n_shards, umgap_index** out) {
    if (!path || !out) return fail(UMGAP_ERR_INVALID, "umgap_index_open: null argument");
    if (n_shards == 0) n_shards = 1;
    if (shard >= n_shards) return fail(UMGAP_ERR_INVALID, "umgap_index_open: shard >= n_shards");
    if (!device_mask) return fail(UMGAP_ERR_INVALID, "umgap_index_open: empty device mask");
    int device = __builtin_ctz(device_mask);
    FILE* f = fopen(path, "rb");
    if (!f) return fail(UMGAP_ERR_IO, std::string("cannot open ") + path + ": " + strerror(errno));
    IndexFileHeader h{};
    if (fread(&h, sizeof(h), 1, f) != 1 || memcmp(h.magic, "UMGAPIX1", 8) != 0) {
        fclose(f);
        return fail(UMGAP_ERR_FORMAT, std::string(path) + ": not a UMGAPIX1 index (fst files must be converted: umgap printindex | umgap buildindex)");
    }
    if (h.version == 1 && h.kind == 2) {
        if (n_shards != 1) { fclose(f); return fail(UMGAP_ERR_INVALID, "variable-length indexes cannot be sharded yet"); }
        umgap_index* vx = nullptr;
        int rcv = new_handle(device, &vx);
        if (!rcv) rcv = var_open(vx, h, f, path);
        fclose(f);
        if (rcv) { if (vx) umgap_index_free(vx); return rcv; }
        *out = vx;
        return UMGAP_OK;
    }
    if (h.version != 1 || (h.kind != 9 && h.kind != 10) || h.r_bits != R_BITS || h.slots != SLOTS || h.mix_c1 != MIX_C1 || h.mix_c2 != MIX_C2 ||
        h.m == 0 || h.nb != ((uint64_t)h.m << A_BITS)) {
        fclose(f);
        return fail(UMGAP_ERR_FORMAT, std::string(path) + ": unsupported UMGAPIX1 parameters");
    }
    if (h.kind == 10 && n_shards != 1) { fclose(f); return fail(UMGAP_ERR_INVALID, "a walk-order index cannot be sharded"); }
    if (h.kind == 10 && ((h.reserved >> 63) != 0 || ((h.reserved >> 60) & 3) > 2 || (h.reserved & ((1ull << 60) - 1)) >= ((0xFFFFFFFFull) << ((h.reserved >> 60) & 3)))) {
        fclose(f);
        return fail(UMGAP_ERR_FORMAT, std::string(path) + ": walk array too long for its entry numbers");
    }
    umgap_index* ix = nullptr;
    int rc = new_handle(device, &ix);
    if (rc) { fclose(f); return rc; }
    uint32_t lo = (uint32_t)((uint64_t)h.nb * shard / n_shards), hi = (uint32_t)((uint64_t)h.nb * (shard + 1) / n_shards);
    uint32_t resident = hi - lo + MAX_DISP;
    hipError_t e = hipMalloc((void**)&ix->d_slots, (size_t)resident * 64);
    if (e != hipSuccess) { fclose(f); umgap_index_free(ix); return fail(UMGAP_ERR_NOMEM, std::string("hipMalloc of the table failed: ") + hipGetErrorString(e)); }
    ix->n_keys = h.n_keys;
    ix->view = TableView{ix->d_slots, h.m, h.nb, lo, resident, lo, hi, nullptr, nullptr, 0};
    // body [lo, hi) then the halo (wraps to bucket 0 for the last shard).  The ~100 GB body is pulled in by several
    // threads, each with its own pinned buffer and stream: pread (page cache or NVMe queue depth) of one chunk overlaps
    // the uploads of the others (a single read-then-copy loop tops out near 4 GB/s).
    // pinned staging per loader thread: 32 MiB for the big tables, less for small ones (pinning memory is not free and
    // the command-line tools open an index per invocation)
    const int fd = fileno(f);
    bool ok = load_file_range(fd, (off_t)(sizeof(h) + (uint64_t)lo * 64), (size_t)(hi - lo) * 64, ix->d_slots, device);
    {
        char halo[MAX_DISP * 64];
        for (uint32_t d = 0; ok && d < MAX_DISP; d++)
            if (pread(fd, halo + d * 64, 64, (off_t)(sizeof(h) + (((uint64_t)hi + d) % h.nb) * 64)) != 64) ok = false;
        if (ok && hipMemcpy((char*)ix->d_slots + (size_t)(hi - lo) * 64, halo, sizeof(halo), hipMemcpyHostToDevice) != hipSuccess) ok = false;
    }
    if (ok && h.n_stash) {
        std::vector<uint64_t> kv(2 * (size_t)h.n_stash);
        ok = fseeko(f, (off_t)(sizeof(h) + (uint64_t)h.nb * 64), SEEK_SET) == 0 && fread(kv.data(), 8, kv.size(), f) == kv.size();
        if (ok) {
            std::vector<std::pair<uint64_t, uint64_t>> entries(h.n_stash);
            for (size_t i = 0; i < h.n_stash; i++) entries[i] = {kv[2 * i], kv[2 * i + 1]};
            if (set_stash(ix, entries)) { fclose(f); umgap_index_free(ix); return UMGAP_ERR_FORMAT; }
        }
    }
    if (ok && h.kind == 10) {   // the walk array behind the stash
        ix->kind = 10;
        ix->n_walk = h.reserved & ((1ull << 60) - 1);
        ix->walk_shift = (uint32_t)((h.reserved >> 60) & 3);
        ix->walk_zero_absent = ((h.reserved >> 62) & 1) != 0;
        if (hipMalloc((void**)&ix->d_walk, std::max<uint64_t>(ix->n_walk, 1) * 8) != hipSuccess) { fclose(f); umgap_index_free(ix); return fail(UMGAP_ERR_NOMEM, "hipMalloc of the walk array failed"); }
        const off_t walk_at = (off_t)(sizeof(h) + (uint64_t)h.nb * 64 + h.n_stash * 16);
        ok = load_file_range(fd, walk_at, (size_t)ix->n_walk * 8, ix->d_walk, device);
        if (ok && ix->walk_shift) {     // the exception list behind the walk array
            uint64_t n_exc = 0;
            ok = fseeko(f, walk_at + (off_t)(ix->n_walk * 8), SEEK_SET) == 0 && fread(&n_exc, 8, 1, f) == 1 && n_exc < (1ull << 26);
            std::vector<uint64_t> k(ok ? n_exc : 0), v(ok ? n_exc : 0);
            if (ok && n_exc) ok = fread(k.data(), 8, n_exc, f) == n_exc && fread(v.data(), 8, n_exc, f) == n_exc;
            if (ok) {
                std::vector<std::pair<uint64_t, uint64_t>> ex(n_exc);
                for (size_t i = 0; i < n_exc; i++) ex[i] = {k[i], v[i]};
                if (const int xrc = set_walk_exceptions(ix, ex)) { fclose(f); umgap_index_free(ix); return xrc; }   // its own code: format, memory or device
            }
        }
    }
    fclose(f);
    if (!ok) { umgap_index_free(ix); return fail(UMGAP_ERR_IO, std::string(path) + ": truncated or unreadable table"); }
    *out = ix;
    return UMGAP_OK;
}

int umgap_index_open(const char* path, uint32_t device_mask, umgap_index** out) {
    return umgap_index_open_shard(path, device_mask, 0, 1, out);
}

int umgap_index_set_shard(umgap_index* ix, uint32_t shard, uint32_t n_shards) {
    if (!ix || n_shards == 0 || shard >= n_shards) return fail(UMGAP_ERR_INVALID, "umgap_index_set_shard: bad argument");
    if (ix->kind != 9) return fail(UMGAP_ERR_INVALID, "umgap_index_set_shard: only the plain 9-mer table is sharded (a walk-order index answers through its walk array, which is not)");
    if (ix->view.first_bucket != 0 || ix->view.n_resident != ix->view.nb + MAX_DISP) return fail(UMGAP_ERR_INVALID, "umgap_index_set_shard: handle does not hold the whole table");
    uint32_t nb = ix->view.nb;
    // the whole table stays resident; only the owned home range shrinks (slots pointer is kept at
    // bucket 0, so "local" indices are offset back by home_lo)
    ix->view.home_lo = (uint32_t)((uint64_t)nb * shard / n_shards);
    ix->view.home_hi = (uint32_t)((uint64_t)nb * (shard + 1) / n_shards);
    ix->view.slots = ix->d_slots + (size_t)ix->view.home_lo * SLOTS;
    return UMGAP_OK;
}

int umgap_index_owner9_dev(const umgap_index* ix, const uint64_t* d_keys, size_t n, uint32_t n_shards, uint32_t* d_owner, void* stream) {
    if (!ix || !n_shards || (n && (!d_keys || !d_owner))) return fail(UMGAP_ERR_INVALID, "umgap_index_owner9_dev: bad argument");
    int rc = use_device(ix->device);
    if (rc) return rc;
    if (n) hipLaunchKernelGGL(k_owner9, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, d_keys, n, ix->view.m, ix->view.nb, n_shards, d_owner);
    UMGAP_HIP(hipGetLastError());
    return UMGAP_OK;
}

int umgap_index_build_windows9_dev(const uint8_t* d_bytes, const uint64_t* d_offsets, const uint32_t* d_taxa, size_t n_fragments,
                                   uint32_t max_fragment_len, uint64_t expected_keys, double load, int device, umgap_index** out) {
    if (!out || (n_fragments && (!d_bytes || !d_offsets || !d_taxa)) || max_fragment_len < 9) return fail(UMGAP_ERR_INVALID, "umgap_index_build_windows9_dev: bad argument");
    umgap_index* ix = nullptr;
    int rc = new_handle(device, &ix);
    if (rc) return rc;
    // no stash here: a chain longer than MAX_DISP grows the table instead (synthetic use only)
    rc = build_with(ix, expected_keys, load, 0, 1, [&](umgap_index* x, BuildFlags* f, uint64_t*) {
        if (n_fragments) hipLaunchKernelGGL(k_insert_windows, dim3(grid_for(n_fragments * (size_t)max_fragment_len, 256, 256 * 16)), dim3(256), 0, x->stream,
                                            x->d_slots, x->view.m, x->view.nb, d_bytes, d_offsets, d_taxa, n_fragments, max_fragment_len, f, (const uint64_t*)nullptr);
        return UMGAP_OK;
    });
    if (rc) { umgap_index_free(ix); return rc; }
    *out = ix;
    return UMGAP_OK;
}

// ---- The 9-mer index in WALK order (kind 10, index_walk.hip), built from a VALUES table (key -> taxon: any kind-9 index, e.g. the
// one `umgap buildindex` makes from the reference's TSV) and a proteome handed over in chunks, in order.  The position table maps a
// key to the number p of its entry E[p] in the walk array; E lists every key ONCE, at the first window of the proteome that holds
// it, in proteome order ("continues" = the entry in front is the previous window of the same protein), and the keys the proteome
// never showed behind them as walks of one.  Keys of the proteome that the values table does not hold get no entry: they miss.
}  // extern "C"
struct umgap_walk_builder {
    const umgap_index* values = nullptr;   // not owned; null: the values are the proteome's own taxa (umgap_index_build_walk9_dev)
    uint64_t expected_keys = 0;            // values == null: an upper bound on the distinct 9-mers
    std::vector<std::pair<uint64_t*, uint64_t>> kept_lists;   // values == null: per chunk, (key, entry) of the kept repeats (device, count)
    uint64_t stash_dups = 0;               // entries walk_finalize_table dropped (a key that went to the stash twice)
    std::vector<std::pair<uint64_t, uint64_t>> dropped;   // ... and which (stash key form, table value): their entries stay in the walk array
    umgap_index* ix = nullptr;             // position table + walk array under construction
    BuildFlags* d_flags = nullptr;
    uint64_t* d_stash = nullptr;
    uint64_t cap_walk = 0, n_walk = 0;
    uint64_t slack = 0, n_kept = 0;        // entries the array may spend on kept repeats (k_walk_keep_short_repeats), and has
    uint32_t shift = 0;                    // the table holds entry number >> shift (walk_ref.hpp)
    uint64_t* d_exc = nullptr;             // (key, position) pairs of the exception list, in the order found
    unsigned long long* d_exc_n = nullptr;
    // deferred values (umgap_walk_builder_new_deferred): no values table at all -- every 9-mer of the proteome gets an entry with
    // value 0 ("no key"), umgap_walk_builder_set_values names the keys afterwards, keys the proteome never showed are appended
    bool deferred = false, sealed = false;
    uint64_t* d_left_keys = nullptr;       // (key, value) pairs set_values did not find an entry for
    uint32_t* d_left_vals = nullptr;
    uint64_t left_cap = 0;
    unsigned long long* d_set_counts = nullptr;   // [0] leftovers, [1] entries valued, [2] keys given two different values
};
constexpr uint64_t WALK_EXC_CAP = 1ull << 22;
namespace {
// the raw stash list of a table under construction -> its sorted stash (a key that went there twice -- see lane_get_wrap -- keeps
// its first entry) and the halo
int walk_finalize_table(umgap_walk_builder* b) {
    umgap_index* x = b->ix;
    BuildFlags h{};
    hipError_t e = hipMemcpyAsync(&h, b->d_flags, sizeof(h), hipMemcpyDeviceToHost, x->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(x->stream);
    if (e != hipSuccess) return fail(UMGAP_ERR_DEVICE, std::string("walk index build: ") + hipGetErrorString(e));
    if (h.bad_key || h.duplicate) return fail(UMGAP_ERR_BUILD, "walk index build: internal error (a key was listed twice)");
    if (h.too_full) return fail(UMGAP_ERR_BUILD, "walk index build: position table too full");
    std::vector<uint64_t> raw(2 * (size_t)h.n_stash);
    if (h.n_stash && hipMemcpy(raw.data(), b->d_stash, raw.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return fail(UMGAP_ERR_DEVICE, "walk index build: stash copy failed");
    std::vector<std::pair<uint64_t, uint64_t>> entries(h.n_stash);
    for (size_t i = 0; i < h.n_stash; i++) entries[i] = {raw[2 * i], raw[2 * i + 1]};
    std::sort(entries.begin(), entries.end());
    b->dropped.clear();
    for (size_t i = 1; i < entries.size(); i++) if (entries[i].first == entries[i - 1].first) b->dropped.push_back(entries[i]);
    entries.erase(std::unique(entries.begin(), entries.end(), [](const std::pair<uint64_t, uint64_t>& p, const std::pair<uint64_t, uint64_t>& q) { return p.first == q.first; }), entries.end());
    b->stash_dups = (uint64_t)h.n_stash - entries.size();
    int rc = set_stash(x, entries);
    if (!rc) rc = finish_halo(x);
    return rc;
}
}  // namespace
extern "C" {

void umgap_walk_builder_free(umgap_walk_builder* b) {
    if (!b) return;
    if (b->ix) { (void)use_device(b->ix->device); (void)hipStreamSynchronize(b->ix->stream); }
    if (b->d_flags) (void)hipFree(b->d_flags);
    if (b->d_stash) (void)hipFree(b->d_stash);
    if (b->d_exc) (void)hipFree(b->d_exc);
    if (b->d_exc_n) (void)hipFree(b->d_exc_n);
    for (void* p : {(void*)b->d_left_keys, (void*)b->d_left_vals, (void*)b->d_set_counts}) if (p) (void)hipFree(p);
    for (auto& kl : b->kept_lists) if (kl.first) (void)hipFree(kl.first);
    if (b->ix) umgap_index_free(b->ix);
    delete b;
}

}  // extern "C"
namespace {
// m = 0: the position table sized for the values table's key count at `load`; otherwise m sub-buckets per superblock
// values == null: no values table -- every 9-mer of the proteome is a key (at most expected_keys of them, on `device`), its value the
// smallest taxon of the fragments that hold it (walk_add_chunk's frag_taxa, k_walk_min_taxa at the end)
int walk_builder_create(const umgap_index* values, double load, uint32_t m, umgap_walk_builder** out, uint64_t expected_keys = 0, int device = 0) {
    if (!out || (!values && !expected_keys)) return fail(UMGAP_ERR_INVALID, "umgap_walk_builder_new: null argument");
    if (values && (values->kind != 9 || values->bound_to || values->view.home_lo != 0 || values->view.home_hi != values->view.nb))
        return fail(UMGAP_ERR_INVALID, "umgap_walk_builder_new: the values must be a whole, unbound 9-mer table");
    const uint64_t n_keys = values ? values->n_keys : expected_keys;
    // entries: every key once, the second entries of keys that went to the stash, kept repeats (slack), padding (shift > 0).  The
    // table's 32-bit values hold the entry number >> shift: shift 0 up to 2^32 - 2^22 entries, 1 up to 2^33, 2 up to 2^34
    // (UMGAP_WALK_SHIFT in the environment asks for at least that shift: tests exercise the wide form on small indexes)
    uint32_t shift = 0;
    if (const char* env = getenv("UMGAP_WALK_SHIFT")) { const int v = atoi(env); if (v >= 0 && v <= 2) shift = (uint32_t)v; }
    uint64_t slack = n_keys / 8, cap = 0;
    for (;; shift++) {
        if (shift > 2) return fail(UMGAP_ERR_INVALID, "umgap_walk_builder_new: more than 2^34 entries");
        const uint64_t pad = shift ? ((1ull << shift) - 1ull) * (n_keys / 16 + 4096) : 0;
        cap = n_keys + STASH_CAP + slack + pad;
        if (cap < ((0xFFFFFFFFull - (1ull << 22)) << shift)) break;
    }
    umgap_walk_builder* b = new umgap_walk_builder();
    b->shift = shift;
    b->values = values;
    b->expected_keys = n_keys;
    int rc = new_handle(values ? values->device : device, &b->ix);
    if (rc) { delete b; return rc; }
    auto bail = [&](int code) { umgap_walk_builder_free(b); return code; };
    rc = alloc_table(b->ix, m ? m : choose_m(std::max<uint64_t>(n_keys, 1), load), 0, 1);
    if (rc) return bail(rc);
    b->slack = slack;
    b->cap_walk = cap;
    if (hipMalloc((void**)&b->d_flags, sizeof(BuildFlags)) != hipSuccess || hipMalloc((void**)&b->d_stash, (size_t)STASH_CAP * 16) != hipSuccess ||
        hipMalloc((void**)&b->ix->d_walk, b->cap_walk * 8) != hipSuccess || hipMalloc((void**)&b->d_exc, WALK_EXC_CAP * 16) != hipSuccess ||
        hipMalloc((void**)&b->d_exc_n, 8) != hipSuccess)
        return bail(fail(UMGAP_ERR_NOMEM, "umgap_walk_builder_new: hipMalloc failed"));
    if (hipMemsetAsync(b->d_exc_n, 0, 8, b->ix->stream) != hipSuccess) return bail(fail(UMGAP_ERR_DEVICE, "umgap_walk_builder_new: memset failed"));
    const size_t nslots = (size_t)b->ix->view.n_resident * SLOTS;
    hipLaunchKernelGGL(k_fill, dim3(grid_for(nslots)), dim3(256), 0, b->ix->stream, b->ix->d_slots, nslots, EMPTY_SLOT);
    if (hipMemsetAsync(b->d_flags, 0, sizeof(BuildFlags), b->ix->stream) != hipSuccess) return bail(fail(UMGAP_ERR_DEVICE, "umgap_walk_builder_new: memset failed"));
    *out = b;
    return UMGAP_OK;
}
}  // namespace
extern "C" {

int umgap_walk_builder_new(const umgap_index* values, double load, umgap_walk_builder** out) { return walk_builder_create(values, load, 0, out); }

int umgap_walk_builder_new_deferred(uint64_t expected_keys, double load, int device, umgap_walk_builder** out) {
    if (!out || !expected_keys) return fail(UMGAP_ERR_INVALID, "umgap_walk_builder_new_deferred: null argument");
    int rc = walk_builder_create(nullptr, load, 0, out, expected_keys, device);
    if (rc) return rc;
    (*out)->deferred = true;
    return UMGAP_OK;
}

}  // extern "C"
namespace {
int walk_install_exceptions(umgap_walk_builder* b);
// the proteome is complete: the table gets its sorted stash and its halo, the walk array its exception list -- look-ups work from here on
int walk_seal(umgap_walk_builder* b) {
    if (b->sealed) return UMGAP_OK;
    umgap_index* x = b->ix;
    int rc = walk_finalize_table(b);
    if (rc) return rc;
    x->walk_shift = b->shift;
    x->n_walk = b->n_walk;
    if ((rc = walk_install_exceptions(b))) return rc;
    b->left_cap = std::max<uint64_t>(b->expected_keys / 16, 1u << 20);     // (12 bytes each, resident until finish)
    if (b->n_walk + b->left_cap > b->cap_walk) b->left_cap = b->cap_walk > b->n_walk ? b->cap_walk - b->n_walk : 0;
    if (hipMalloc((void**)&b->d_left_keys, std::max<uint64_t>(b->left_cap, 1) * 8) != hipSuccess || hipMalloc((void**)&b->d_left_vals, std::max<uint64_t>(b->left_cap, 1) * 4) != hipSuccess ||
        hipMalloc((void**)&b->d_set_counts, 24) != hipSuccess)
        return fail(UMGAP_ERR_NOMEM, "umgap_walk_builder_set_values: hipMalloc failed");
    UMGAP_HIP(hipMemset(b->d_set_counts, 0, 24));
    b->sealed = true;
    return UMGAP_OK;
}
}  // namespace
extern "C" {

int umgap_walk_builder_set_values_dev(umgap_walk_builder* b, const uint64_t* d_keys45, const uint32_t* d_values, size_t n) {
    if (!b || !b->ix || (n && (!d_keys45 || !d_values))) return fail(UMGAP_ERR_INVALID, "umgap_walk_builder_set_values_dev: null argument");
    if (!b->deferred) return fail(UMGAP_ERR_INVALID, "umgap_walk_builder_set_values_dev: this builder was given its values as a table");
    int rc = use_device(b->ix->device);
    if (rc) return rc;
    UMGAP_HIP(hipDeviceSynchronize());   // the caller's arrays may have been produced on any stream
    if ((rc = walk_seal(b))) return rc;
    umgap_index* x = b->ix;
    if (n) hipLaunchKernelGGL(k_walk_set_values, dim3(grid_for(n)), dim3(256), 0, x->stream, x->view, walk_ref_of(x), d_keys45, d_values, n, b->d_left_keys, b->d_left_vals, b->left_cap,
                              b->d_set_counts);
    UMGAP_HIP(hipGetLastError());
    UMGAP_HIP(hipStreamSynchronize(x->stream));
    return UMGAP_OK;
}

int umgap_walk_builder_set_values(umgap_walk_builder* b, const uint64_t* keys45, const uint32_t* values, size_t n) {
    if (!b || !b->ix || (n && (!keys45 || !values))) return fail(UMGAP_ERR_INVALID, "umgap_walk_builder_set_values: null argument");
    int rc = use_device(b->ix->device);
    if (rc) return rc;
    const size_t step = 64u << 20;
    uint64_t* d_k = nullptr;
    uint32_t* d_v = nullptr;
    if (hipMalloc((void**)&d_k, std::min(std::max<size_t>(n, 1), step) * 8) != hipSuccess || hipMalloc((void**)&d_v, std::min(std::max<size_t>(n, 1), step) * 4) != hipSuccess) {
        if (d_k) (void)hipFree(d_k);
        return fail(UMGAP_ERR_NOMEM, "umgap_walk_builder_set_values: hipMalloc failed");
    }
    rc = n ? UMGAP_OK : umgap_walk_builder_set_values_dev(b, d_k, d_v, 0);      // (n == 0 still seals the builder)
    for (size_t o = 0; o < n && !rc; o += step) {
        const size_t len = std::min(step, n - o);
        if (hipMemcpy(d_k, keys45 + o, len * 8, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(d_v, values + o, len * 4, hipMemcpyHostToDevice) != hipSuccess)
            rc = fail(UMGAP_ERR_DEVICE, "umgap_walk_builder_set_values: copy failed");
        if (!rc) rc = umgap_walk_builder_set_values_dev(b, d_k, d_v, len);
    }
    (void)hipFree(d_k);
    (void)hipFree(d_v);
    return rc;
}

}  // extern "C"
namespace {
int walk_add_chunk(umgap_walk_builder* b, const uint8_t* d_bytes, const uint64_t* d_offsets, size_t n_fragments, uint32_t max_fragment_len, const uint32_t* d_frag_taxa);
}
extern "C" {
// the next chunk of the proteome (fragments in order; d_offsets[0 .. n_fragments] index d_bytes; fewer than 2^32 - 1 windows per chunk)
int umgap_walk_builder_add_dev(umgap_walk_builder* b, const uint8_t* d_bytes, const uint64_t* d_offsets, size_t n_fragments, uint32_t max_fragment_len) {
    if (b && !b->values && !b->deferred) return fail(UMGAP_ERR_INVALID, "umgap_walk_builder_add_dev: this builder takes its values from the proteome (internal)");
    if (b && b->deferred) {
        if (b->sealed) return fail(UMGAP_ERR_INVALID, "umgap_walk_builder_add_dev: the proteome is complete once values are being set");
        if (!n_fragments) return UMGAP_OK;
        // every entry starts as "no key": a fragment taxon of 0 for all (the passes then skip their values look-up, as for a proteome's own index)
        int rc = use_device(b->ix->device);
        if (rc) return rc;
        uint32_t* d_zero = nullptr;
        if (hipMalloc((void**)&d_zero, n_fragments * 4) != hipSuccess) return fail(UMGAP_ERR_NOMEM, "umgap_walk_builder_add_dev: hipMalloc failed");
        if (hipMemset(d_zero, 0, n_fragments * 4) != hipSuccess) { (void)hipFree(d_zero); return fail(UMGAP_ERR_DEVICE, "umgap_walk_builder_add_dev: memset failed"); }
        rc = walk_add_chunk(b, d_bytes, d_offsets, n_fragments, max_fragment_len, d_zero);
        (void)hipDeviceSynchronize();
        (void)hipFree(d_zero);
        return rc;
    }
    return walk_add_chunk(b, d_bytes, d_offsets, n_fragments, max_fragment_len, nullptr);
}
}  // extern "C"
namespace {
// d_frag_taxa (builders without a values table): the taxon of every fragment of the chunk
int walk_add_chunk(umgap_walk_builder* b, const uint8_t* d_bytes, const uint64_t* d_offsets, size_t n_fragments, uint32_t max_fragment_len, const uint32_t* d_frag_taxa) {
    if (!b || !b->ix || (n_fragments && (!d_bytes || !d_offsets)) || (!b->values && !d_frag_taxa)) return fail(UMGAP_ERR_INVALID, "umgap_walk_builder_add_dev: null argument");
    if (!n_fragments || max_fragment_len < 9) return UMGAP_OK;
    umgap_index* x = b->ix;
    int rc = use_device(x->device);
    if (rc) return rc;
    UMGAP_HIP(hipDeviceSynchronize());   // the caller's arrays are read on the builder's stream
    umgap_index* c = nullptr;
    uint32_t* d_cnt = nullptr;
    uint8_t* d_flag = nullptr;
    uint64_t *d_wbase = nullptr, *d_fbase = nullptr, *d_total = nullptr;
    auto done = [&](int code) {
        for (void* p : {(void*)d_cnt, (void*)d_flag, (void*)d_wbase, (void*)d_fbase, (void*)d_total}) if (p) (void)hipFree(p);
        if (c) umgap_index_free(c);
        return code;
    };
    if (hipMalloc((void**)&d_cnt, n_fragments * 4) != hipSuccess || hipMalloc((void**)&d_wbase, (n_fragments + 1) * 8) != hipSuccess ||
        hipMalloc((void**)&d_fbase, (n_fragments + 1) * 8) != hipSuccess || hipMalloc((void**)&d_total, 24) != hipSuccess)
        return done(fail(UMGAP_ERR_NOMEM, "umgap_walk_builder_add_dev: hipMalloc failed"));
    // the chunk's windows numbered fragment by fragment
    hipLaunchKernelGGL(k_window_counts, dim3(grid_for(n_fragments)), dim3(256), 0, x->stream, d_offsets, n_fragments, d_cnt);
    hipLaunchKernelGGL(k_scan_counts, dim3(1), dim3(1024), 0, x->stream, d_cnt, n_fragments, d_wbase, d_total);
    uint64_t n_local = 0;
    hipError_t e = hipMemcpyAsync(&n_local, d_total, 8, hipMemcpyDeviceToHost, x->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(x->stream);
    if (e != hipSuccess) return done(fail(UMGAP_ERR_DEVICE, std::string("umgap_walk_builder_add_dev: ") + hipGetErrorString(e)));
    if (n_local >= 0xFFFFFFFFull) return done(fail(UMGAP_ERR_INVALID, "umgap_walk_builder_add_dev: a chunk must hold fewer than 2^32 - 1 windows"));
    if (!n_local) return done(UMGAP_OK);
    // C: the chunk's candidates (new keys of the index -> their first window in the chunk)
    rc = new_handle(x->device, &c);
    if (rc) return done(rc);
    const size_t iter = n_fragments * (size_t)max_fragment_len;
    const TableView V = b->values ? b->values->view : TableView{};
    rc = build_with(c, std::min<uint64_t>(n_local, std::max<uint64_t>(b->expected_keys, 1)), 0.5, 0, 1, [&](umgap_index* t, BuildFlags* f, uint64_t*) {
        hipLaunchKernelGGL(k_walk_candidates, dim3(grid_for(iter, 256, 256 * 16)), dim3(256), 0, t->stream, t->d_slots, t->view.m, t->view.nb, d_bytes, d_offsets,
                           (const uint64_t*)d_wbase, n_fragments, max_fragment_len, f, V, (const uint64_t*)x->d_slots, x->view.m, x->view.nb, d_frag_taxa);
        return UMGAP_OK;
    });
    if (rc) return done(rc);
    if (hipMalloc((void**)&d_flag, n_local) != hipSuccess) return done(fail(UMGAP_ERR_NOMEM, "umgap_walk_builder_add_dev: hipMalloc failed"));
    // a window beyond max_fragment_len is never visited by the flag pass: it must read as "no entry" (its key then follows as a
    // walk of one), not as whatever the allocation held
    if ((e = hipMemsetAsync(d_flag, 0, n_local, x->stream)) != hipSuccess) return done(fail(UMGAP_ERR_DEVICE, std::string("umgap_walk_builder_add_dev: ") + hipGetErrorString(e)));
    hipLaunchKernelGGL(k_walk_flags, dim3(grid_for(iter, 256, 256 * 16)), dim3(256), 0, x->stream, c->view, V, d_bytes, d_offsets, (const uint64_t*)d_wbase, n_fragments,
                       max_fragment_len, d_flag, d_frag_taxa);
    uint64_t n_new = 0, n_keep = 0;
    for (uint32_t gap : {WALK_KEEP_REPEATS, 0u}) {   // short runs of repeats keep their entries while the array has room for them
        if ((e = hipMemsetAsync(d_total + 2, 0, 8, x->stream)) != hipSuccess) return done(fail(UMGAP_ERR_DEVICE, std::string("umgap_walk_builder_add_dev: ") + hipGetErrorString(e)));
        hipLaunchKernelGGL(k_walk_keep_short_repeats, dim3(grid_for(n_fragments)), dim3(256), 0, x->stream, d_flag, (const uint64_t*)d_wbase, n_fragments, gap, d_cnt,
                           (unsigned long long*)(d_total + 2), 1u << b->shift);
        hipLaunchKernelGGL(k_scan_counts, dim3(1), dim3(1024), 0, x->stream, d_cnt, n_fragments, d_fbase, d_total + 1);
        uint64_t two[2] = {0, 0};
        e = hipMemcpyAsync(two, d_total + 1, 16, hipMemcpyDeviceToHost, x->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(x->stream);
        if (e != hipSuccess) return done(fail(UMGAP_ERR_DEVICE, std::string("umgap_walk_builder_add_dev: ") + hipGetErrorString(e)));
        n_new = two[0];
        n_keep = two[1];
        if (b->n_kept + n_keep <= b->slack) break;
    }
    b->n_kept += n_keep;
    if (b->n_walk + n_new > b->cap_walk) return done(fail(UMGAP_ERR_BUILD, "umgap_walk_builder_add_dev: the walk array is full (more fragments than one per 16 keys pad their runs: internal limit)"));
    uint64_t* d_kept = nullptr;       // without a values table: where the chunk's kept repeats went (their taxa are settled at the end)
    if (d_frag_taxa && n_keep) {
        if (hipMalloc((void**)&d_kept, n_keep * 16) != hipSuccess) return done(fail(UMGAP_ERR_NOMEM, "umgap_index_build_walk9_dev: hipMalloc failed"));
        b->kept_lists.emplace_back(d_kept, n_keep);
        if ((e = hipMemsetAsync(&b->d_flags->n_kept_noted, 0, 8, x->stream)) != hipSuccess) return done(fail(UMGAP_ERR_DEVICE, std::string("umgap_walk_builder_add_dev: ") + hipGetErrorString(e)));
    }
    if (n_new)
        hipLaunchKernelGGL(k_walk_emit, dim3(grid_for(n_fragments)), dim3(256), 0, x->stream, d_bytes, d_offsets, (const uint64_t*)d_wbase, (const uint64_t*)d_fbase,
                           (const uint8_t*)d_flag, n_fragments, V, x->d_walk, b->n_walk, x->d_slots, x->view.m, x->view.nb, b->d_flags, b->d_stash, b->shift, b->d_exc,
                           b->d_exc_n, WALK_EXC_CAP, d_frag_taxa, d_kept, (uint64_t)n_keep);
    e = hipStreamSynchronize(x->stream);
    if (e != hipSuccess) return done(fail(UMGAP_ERR_DEVICE, std::string("umgap_walk_builder_add_dev: ") + hipGetErrorString(e)));
    b->n_walk += n_new;
    return done(UMGAP_OK);
}
}  // namespace
extern "C" {

// the same from host memory (the chunk is copied to the device for the call)
int umgap_walk_builder_add(umgap_walk_builder* b, const uint8_t* bytes, const uint64_t* offsets, size_t n_fragments, uint32_t max_fragment_len) {
    if (!b || !b->ix || (n_fragments && (!bytes || !offsets))) return fail(UMGAP_ERR_INVALID, "umgap_walk_builder_add: null argument");
    if (!n_fragments) return UMGAP_OK;
    int rc = use_device(b->ix->device);
    if (rc) return rc;
    const uint64_t base = offsets[0], total = offsets[n_fragments] - base;
    std::vector<uint64_t> rel(n_fragments + 1);
    for (size_t i = 0; i <= n_fragments; i++) {
        if (offsets[i] < base || (i && offsets[i] < offsets[i - 1])) return fail(UMGAP_ERR_INVALID, "umgap_walk_builder_add: offsets must not decrease");
        rel[i] = offsets[i] - base;
    }
    uint8_t* d_bytes = nullptr;
    uint64_t* d_off = nullptr;
    if (hipMalloc((void**)&d_bytes, total + 16) != hipSuccess || hipMalloc((void**)&d_off, rel.size() * 8) != hipSuccess) {
        if (d_bytes) (void)hipFree(d_bytes);
        return fail(UMGAP_ERR_NOMEM, "umgap_walk_builder_add: hipMalloc failed");
    }
    if ((total && hipMemcpy(d_bytes, bytes + base, total, hipMemcpyHostToDevice) != hipSuccess) || hipMemcpy(d_off, rel.data(), rel.size() * 8, hipMemcpyHostToDevice) != hipSuccess)
        rc = fail(UMGAP_ERR_DEVICE, "umgap_walk_builder_add: copy failed");
    if (!rc) rc = umgap_walk_builder_add_dev(b, d_bytes, d_off, n_fragments, max_fragment_len);
    (void)hipFree(d_bytes);
    (void)hipFree(d_off);
    return rc;
}

}  // extern "C"
namespace {
int walk_finish(umgap_walk_builder* b, umgap_index** out, std::vector<std::pair<uint64_t, uint64_t>>* dropped_out);
// the builder's exception pairs so far -> the index's sorted exception list (walk_ref.hpp)
int walk_install_exceptions(umgap_walk_builder* b) {
    umgap_index* x = b->ix;
    if (!b->shift) return UMGAP_OK;
    unsigned long long n_exc = 0;
    if (hipMemcpy(&n_exc, b->d_exc_n, 8, hipMemcpyDeviceToHost) != hipSuccess) return fail(UMGAP_ERR_DEVICE, "umgap_walk_builder_finish: copy failed");
    if (n_exc > WALK_EXC_CAP) return fail(UMGAP_ERR_BUILD, "umgap_walk_builder_finish: more than 2^22 keys share their low bits with a neighbour (low-complexity proteome?)");
    std::vector<uint64_t> raw(2 * (size_t)n_exc);
    if (n_exc && hipMemcpy(raw.data(), b->d_exc, raw.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return fail(UMGAP_ERR_DEVICE, "umgap_walk_builder_finish: copy failed");
    std::vector<std::pair<uint64_t, uint64_t>> ex(n_exc);
    for (size_t i = 0; i < n_exc; i++) ex[i] = {raw[2 * i], raw[2 * i + 1]};
    std::sort(ex.begin(), ex.end());
    ex.erase(std::unique(ex.begin(), ex.end(), [](const std::pair<uint64_t, uint64_t>& p, const std::pair<uint64_t, uint64_t>& q) { return p.first == q.first; }), ex.end());
    return set_walk_exceptions(x, ex);
}
}
extern "C" {
int umgap_walk_builder_finish(umgap_walk_builder* b, umgap_index** out) { return walk_finish(b, out, nullptr); }
}  // extern "C"
namespace {
// dropped_out: receives the stash entries that were dropped as duplicates (builders without a values table settle their taxa)
int walk_finish(umgap_walk_builder* b, umgap_index** out, std::vector<std::pair<uint64_t, uint64_t>>* dropped_out) {
    if (!b || !b->ix || !out) return fail(UMGAP_ERR_INVALID, "umgap_walk_builder_finish: null argument");
    umgap_index* x = b->ix;
    int rc = use_device(x->device);
    if (rc) return rc;
    if (b->deferred) rc = walk_seal(b);            // (nobody set a value: an index of misses, but a consistent one)
    else rc = walk_finalize_table(b);
    if (rc) return rc;
    if (dropped_out) *dropped_out = b->dropped;
    // the keys the proteome never showed: appended as walks of one
    unsigned long long* d_n = nullptr;
    uint64_t* d_keys = nullptr;
    uint32_t *d_taxa = nullptr, *d_pos = nullptr;
    auto done = [&](int code) {
        for (void* p : {(void*)d_n, (void*)d_keys, (void*)d_taxa, (void*)d_pos}) if (p) (void)hipFree(p);
        return code;
    };
    if (hipMalloc((void**)&d_n, 8) != hipSuccess) return done(fail(UMGAP_ERR_NOMEM, "umgap_walk_builder_finish: hipMalloc failed"));
    const TableView V = b->values ? b->values->view : TableView{};
    const uint64_t inv1 = inv_odd45(MIX_C1), inv2 = inv_odd45(MIX_C2);
    const size_t scan = (size_t)V.nb * SLOTS + V.n_stash;
    unsigned long long n_left = 0;
    hipError_t e = hipSuccess;
    if (b->values) {        // (without a values table there is nothing the proteome did not show)
        UMGAP_HIP(hipMemsetAsync(d_n, 0, 8, x->stream));
        hipLaunchKernelGGL(k_walk_leftovers, dim3(grid_for(scan, 256, 256 * 16)), dim3(256), 0, x->stream, V, x->view, inv1, inv2, (uint64_t*)nullptr, (uint32_t*)nullptr, d_n);
        e = hipMemcpyAsync(&n_left, d_n, 8, hipMemcpyDeviceToHost, x->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(x->stream);
        if (e != hipSuccess) return done(fail(UMGAP_ERR_DEVICE, std::string("umgap_walk_builder_finish: ") + hipGetErrorString(e)));
    }
    unsigned long long set_counts[3] = {0, 0, 0};
    if (b->deferred) {      // what set_values found no entry for
        if (hipMemcpy(set_counts, b->d_set_counts, 24, hipMemcpyDeviceToHost) != hipSuccess) return done(fail(UMGAP_ERR_DEVICE, "umgap_walk_builder_finish: copy failed"));
        if (set_counts[2]) return done(fail(UMGAP_ERR_BUILD, "umgap_walk_builder_finish: " + std::to_string(set_counts[2]) + " keys were given two different values"));
        if (set_counts[0] > b->left_cap)
            return done(fail(UMGAP_ERR_BUILD, "umgap_walk_builder_finish: " + std::to_string(set_counts[0]) + " keys are not 9-mers of the proteome (room for " + std::to_string(b->left_cap) +
                                                  "): a walk-order index wants the proteome its keys come from"));
        n_left = set_counts[0];
    }
    if (n_left) {
        if (b->n_walk + n_left > b->cap_walk) return done(fail(UMGAP_ERR_BUILD, "umgap_walk_builder_finish: more entries than the values table has keys (internal error)"));
        if (b->deferred) { d_keys = b->d_left_keys; d_taxa = b->d_left_vals; b->d_left_keys = nullptr; b->d_left_vals = nullptr; }     // (freed with the rest below)
        if ((!b->deferred && (hipMalloc((void**)&d_keys, n_left * 8) != hipSuccess || hipMalloc((void**)&d_taxa, n_left * 4) != hipSuccess)) || hipMalloc((void**)&d_pos, n_left * 4) != hipSuccess)
            return done(fail(UMGAP_ERR_NOMEM, "umgap_walk_builder_finish: hipMalloc failed"));
        UMGAP_HIP(hipMemsetAsync(d_n, 0, 8, x->stream));
        if (!b->deferred) hipLaunchKernelGGL(k_walk_leftovers, dim3(grid_for(scan, 256, 256 * 16)), dim3(256), 0, x->stream, V, x->view, inv1, inv2, d_keys, d_taxa, d_n);
        // the slots came from an atomic counter: sorted by key, so that the same input gives the same file, byte for byte
        {
            uint64_t* d_keys2 = nullptr;
            uint32_t* d_taxa2 = nullptr;
            void* d_tmp = nullptr;
            size_t tmp_bytes = 0;
            auto drop = [&]() { for (void* p : {(void*)d_keys2, (void*)d_taxa2, d_tmp}) if (p) (void)hipFree(p); };
            if (hipMalloc((void**)&d_keys2, n_left * 8) != hipSuccess || hipMalloc((void**)&d_taxa2, n_left * 4) != hipSuccess) { drop(); return done(fail(UMGAP_ERR_NOMEM, "umgap_walk_builder_finish: hipMalloc failed")); }
            hipError_t se = rocprim::radix_sort_pairs(nullptr, tmp_bytes, d_keys, d_keys2, d_taxa, d_taxa2, (size_t)n_left, 0u, 45u, x->stream);
            if (se == hipSuccess) se = hipMalloc(&d_tmp, tmp_bytes ? tmp_bytes : 8);
            if (se == hipSuccess) se = rocprim::radix_sort_pairs(d_tmp, tmp_bytes, d_keys, d_keys2, d_taxa, d_taxa2, (size_t)n_left, 0u, 45u, x->stream);
            if (se == hipSuccess) se = hipStreamSynchronize(x->stream);
            if (se != hipSuccess) { drop(); return done(fail(UMGAP_ERR_DEVICE, std::string("umgap_walk_builder_finish: sort: ") + hipGetErrorString(se))); }
            std::swap(d_keys, d_keys2);
            std::swap(d_taxa, d_taxa2);
            if (b->deferred) {
                // the same pair may have come twice (a chunk handed over again): one entry per key; two values for a key are an error
                hipLaunchKernelGGL(k_walk_left_conflicts, dim3(grid_for(n_left)), dim3(256), 0, x->stream, (const uint64_t*)d_keys, (const uint32_t*)d_taxa, (size_t)n_left, b->d_set_counts + 2);
                unsigned long long* d_cnt = d_n;
                size_t ub = 0;
                if (d_tmp) { (void)hipFree(d_tmp); d_tmp = nullptr; }
                se = rocprim::unique_by_key(nullptr, ub, d_keys, d_taxa, d_keys2, d_taxa2, d_cnt, (size_t)n_left, rocprim::equal_to<uint64_t>(), x->stream);
                if (se == hipSuccess) se = hipMalloc(&d_tmp, ub ? ub : 8);
                if (se == hipSuccess) se = rocprim::unique_by_key(d_tmp, ub, d_keys, d_taxa, d_keys2, d_taxa2, d_cnt, (size_t)n_left, rocprim::equal_to<uint64_t>(), x->stream);
                unsigned long long uniq = 0, conflicts = 0;
                if (se == hipSuccess) se = hipMemcpyAsync(&uniq, d_cnt, 8, hipMemcpyDeviceToHost, x->stream);
                if (se == hipSuccess) se = hipMemcpyAsync(&conflicts, b->d_set_counts + 2, 8, hipMemcpyDeviceToHost, x->stream);
                if (se == hipSuccess) se = hipStreamSynchronize(x->stream);
                if (se != hipSuccess) { drop(); return done(fail(UMGAP_ERR_DEVICE, std::string("umgap_walk_builder_finish: unique: ") + hipGetErrorString(se))); }
                if (conflicts) { drop(); return done(fail(UMGAP_ERR_BUILD, "umgap_walk_builder_finish: " + std::to_string(conflicts) + " keys were given two different values")); }
                std::swap(d_keys, d_keys2);
                std::swap(d_taxa, d_taxa2);
                set_counts[0] = uniq;
                n_left = uniq;
            }
            drop();
        }
        hipLaunchKernelGGL(k_walk_append, dim3(grid_for(n_left)), dim3(256), 0, x->stream, (const uint64_t*)d_keys, (const uint32_t*)d_taxa, (size_t)n_left, x->d_walk, b->n_walk, d_pos,
                           b->shift, b->d_exc, b->d_exc_n, WALK_EXC_CAP);
        hipLaunchKernelGGL(k_insert, dim3(grid_for(n_left)), dim3(256), 0, x->stream, x->d_slots, x->view.m, x->view.nb, 0u, x->view.nb, (const uint64_t*)d_keys,
                           (const uint32_t*)d_pos, (size_t)n_left, b->d_flags, b->d_stash);
        b->n_walk += n_left;
        rc = walk_finalize_table(b);   // the stash list and the halo again, now with the appended keys
        if (rc) return done(rc);
    }
    // the exception list (walk_ref.hpp): sorted by key, one entry per key
    x->walk_shift = b->shift;
    x->n_walk = b->n_walk;            // (set_walk_exceptions checks the listed positions against it)
    if ((rc = walk_install_exceptions(b))) return done(rc);
    x->kind = 10;
    x->n_walk = b->n_walk;
    x->n_keys = b->values ? b->values->n_keys : b->expected_keys;     // (as announced, like umgap_index_build_windows9_dev's own count)
    if (b->deferred) {
        // entries nothing points at -- kept repeats, the second entries of keys a full chain sent to the stash twice -- take the value of
        // their key's own entry (as the proteome's own builder settles them)
        x->walk_zero_absent = true;
        x->n_keys = set_counts[1] + set_counts[0];
        const WalkRef W = walk_ref_of(x);
        for (auto& kl : b->kept_lists)
            hipLaunchKernelGGL(k_walk_settle_kept, dim3(grid_for(kl.second)), dim3(256), 0, x->stream, x->view, W, (const uint64_t*)kl.first, (size_t)kl.second);
        uint64_t* d_dropped = nullptr;
        hipError_t se = hipSuccess;
        if (!b->dropped.empty()) {
            std::vector<uint64_t> raw(2 * b->dropped.size());
            for (size_t i = 0; i < b->dropped.size(); i++) { raw[2 * i] = b->dropped[i].first; raw[2 * i + 1] = b->dropped[i].second; }
            se = hipMalloc((void**)&d_dropped, raw.size() * 8);
            if (se == hipSuccess) se = hipMemcpy(d_dropped, raw.data(), raw.size() * 8, hipMemcpyHostToDevice);
            if (se == hipSuccess) hipLaunchKernelGGL(k_walk_settle_dropped, dim3(grid_for(b->dropped.size())), dim3(256), 0, x->stream, x->view, W, (const uint64_t*)d_dropped, b->dropped.size(),
                                                     inv_odd45(MIX_C1), inv_odd45(MIX_C2));
        }
        if (se == hipSuccess) se = hipGetLastError();
        if (se == hipSuccess) se = hipStreamSynchronize(x->stream);
        if (d_dropped) (void)hipFree(d_dropped);
        if (se != hipSuccess) return done(fail(UMGAP_ERR_DEVICE, std::string("umgap_walk_builder_finish: ") + hipGetErrorString(se)));
    }
    *out = x;
    b->ix = nullptr;
    umgap_walk_builder_free(b);
    return done(UMGAP_OK);
}
}  // namespace
extern "C" {

// A proteome's own index in walk order: values as umgap_index_build_windows9_dev (the smallest taxon over the fragments that hold
// a 9-mer), the proteome handed to the builder in chunks of at most 2^29 windows (the chunk's temporary table stays small).
int umgap_index_build_walk9_dev(const uint8_t* d_bytes, const uint64_t* d_offsets, const uint32_t* d_taxa, size_t n_fragments,
                                uint32_t max_fragment_len, uint64_t expected_keys, double load, int device, umgap_index** out) {
    if (!out || !n_fragments || !d_bytes || !d_offsets || !d_taxa || max_fragment_len < 9) return fail(UMGAP_ERR_INVALID, "umgap_index_build_walk9_dev: bad argument");
    // Round 4: no values table beside the index under construction (that table -- 164 GB for 8.4 G keys -- was what kept an index of the
    // real size class from being built on one GPU).  Every 9-mer of the proteome is a key; an entry is written with the taxon of the
    // fragment that lists it, and a last pass over the proteome lowers it to the smallest taxon of all fragments that hold the key.
    int rc = use_device(device);
    if (rc) return rc;
    umgap_walk_builder* b = nullptr;
    uint32_t* d_cnt = nullptr;
    uint64_t *d_wbase = nullptr, *d_cuts = nullptr;
    umgap_index* ix = nullptr;
    auto done = [&](int code) {
        for (void* p : {(void*)d_cnt, (void*)d_wbase, (void*)d_cuts}) if (p) (void)hipFree(p);
        if (b) umgap_walk_builder_free(b);
        if (code && ix) umgap_index_free(ix);
        return code;
    };
    UMGAP_HIP(hipDeviceSynchronize());   // the caller's arrays may have been produced on any stream
    if (!expected_keys) expected_keys = (uint64_t)n_fragments * (max_fragment_len - 8);
    rc = walk_builder_create(nullptr, load, 0, &b, expected_keys, device);
    if (rc) return done(rc);
    hipStream_t s = b->ix->stream;
    if (hipMalloc((void**)&d_cnt, n_fragments * 4) != hipSuccess || hipMalloc((void**)&d_wbase, (n_fragments + 2) * 8) != hipSuccess)
        return done(fail(UMGAP_ERR_NOMEM, "umgap_index_build_walk9_dev: hipMalloc failed"));
    hipLaunchKernelGGL(k_window_counts, dim3(grid_for(n_fragments)), dim3(256), 0, s, d_offsets, n_fragments, d_cnt);
    hipLaunchKernelGGL(k_scan_counts, dim3(1), dim3(1024), 0, s, d_cnt, n_fragments, d_wbase, d_wbase + n_fragments + 1);
    uint64_t n_windows = 0;
    hipError_t e = hipMemcpyAsync(&n_windows, d_wbase + n_fragments + 1, 8, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return done(fail(UMGAP_ERR_DEVICE, std::string("umgap_index_build_walk9_dev: ") + hipGetErrorString(e)));
    const uint64_t per = 1ull << 29;
    const uint32_t n_chunks = (uint32_t)std::max<uint64_t>(1, (n_windows + per - 1) / per);
    std::vector<uint64_t> cuts(n_chunks + 1, 0);
    if (hipMalloc((void**)&d_cuts, (n_chunks + 1) * 8) != hipSuccess) return done(fail(UMGAP_ERR_NOMEM, "umgap_index_build_walk9_dev: hipMalloc failed"));
    hipLaunchKernelGGL(k_walk_cuts, dim3((n_chunks + 255) / 256), dim3(256), 0, s, (const uint64_t*)d_wbase, n_fragments, per, n_chunks, d_cuts);
    e = hipMemcpyAsync(cuts.data(), d_cuts, n_chunks * 8, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return done(fail(UMGAP_ERR_DEVICE, std::string("umgap_index_build_walk9_dev: ") + hipGetErrorString(e)));
    cuts[n_chunks] = n_fragments;
    for (void** p : {(void**)&d_cnt, (void**)&d_wbase, (void**)&d_cuts}) { (void)hipFree(*p); *p = nullptr; }
    for (uint32_t c = 0; c < n_chunks && !rc; c++)
        if (cuts[c + 1] > cuts[c]) rc = walk_add_chunk(b, d_bytes, d_offsets + cuts[c], (size_t)(cuts[c + 1] - cuts[c]), max_fragment_len, d_taxa + cuts[c]);
    if (rc) return done(rc);
    std::vector<std::pair<uint64_t*, uint64_t>> kept;
    kept.swap(b->kept_lists);            // (finish frees the builder)
    std::vector<std::pair<uint64_t, uint64_t>> dropped;
    auto drop_kept = [&]() { for (auto& kl : kept) if (kl.first) (void)hipFree(kl.first); kept.clear(); };
    rc = walk_finish(b, &ix, &dropped);
    if (rc) { drop_kept(); return done(rc); }
    b = nullptr;   // consumed
    // the values: every window of the proteome lowers its key's entry to its fragment's taxon; then the kept repeats copy theirs
    s = ix->stream;
    const WalkRef W = walk_ref_of(ix);
    hipLaunchKernelGGL(k_walk_min_taxa, dim3(grid_for(n_fragments * (size_t)max_fragment_len, 256, 256 * 16)), dim3(256), 0, s, ix->view, W, d_bytes, d_offsets, d_taxa, n_fragments,
                       max_fragment_len);
    for (auto& kl : kept)
        hipLaunchKernelGGL(k_walk_settle_kept, dim3(grid_for(kl.second)), dim3(256), 0, s, ix->view, W, (const uint64_t*)kl.first, (size_t)kl.second);
    uint64_t* d_dropped = nullptr;
    if (!dropped.empty()) {
        std::vector<uint64_t> raw(2 * dropped.size());
        for (size_t i = 0; i < dropped.size(); i++) { raw[2 * i] = dropped[i].first; raw[2 * i + 1] = dropped[i].second; }
        if (hipMalloc((void**)&d_dropped, raw.size() * 8) != hipSuccess || hipMemcpyAsync(d_dropped, raw.data(), raw.size() * 8, hipMemcpyHostToDevice, s) != hipSuccess) {
            if (d_dropped) (void)hipFree(d_dropped);
            drop_kept();
            return done(fail(UMGAP_ERR_NOMEM, "umgap_index_build_walk9_dev: hipMalloc failed"));
        }
        hipLaunchKernelGGL(k_walk_settle_dropped, dim3(grid_for(dropped.size())), dim3(256), 0, s, ix->view, W, (const uint64_t*)d_dropped, dropped.size(), inv_odd45(MIX_C1), inv_odd45(MIX_C2));
        e = hipStreamSynchronize(s);      // (raw is read by the copy)
    }
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (d_dropped) (void)hipFree(d_dropped);
    drop_kept();
    if (e != hipSuccess) return done(fail(UMGAP_ERR_DEVICE, std::string("umgap_index_build_walk9_dev: ") + hipGetErrorString(e)));
    *out = ix;
    return done(UMGAP_OK);
}

int umgap_index_partition9_dev(const umgap_index* ix, const uint64_t* d_keys, size_t n, uint32_t n_shards, uint64_t cap,
                               uint64_t* d_send, int32_t* d_perm, uint64_t* d_cursor, uint32_t* d_overflow, void* stream) {
    if ((uint64_t)n_shards * cap > 0x7FFFFFFFull) return fail(UMGAP_ERR_INVALID, "umgap_index_partition9_dev: n_shards * cap must fit 31 bits (cut the batch into pieces)");
    if (!ix || !n_shards || n_shards > 64 || (n && (!d_keys || !d_send || !d_perm)) || !d_cursor || !d_overflow) return fail(UMGAP_ERR_INVALID, "umgap_index_partition9_dev: bad argument");
    int rc = use_device(ix->device);
    if (rc) return rc;
    if (n) hipLaunchKernelGGL(k_partition, dim3(grid_for((n + PART_PER_THREAD - 1) / PART_PER_THREAD)), dim3(256), 0, (hipStream_t)stream, d_keys, n, ix->view.m, ix->view.nb, n_shards, cap,
                              d_send, d_perm, (unsigned long long*)d_cursor, d_overflow);
    UMGAP_HIP(hipGetLastError());
    return UMGAP_OK;
}

int umgap_index_lookup9_dev(umgap_index* ix, const uint64_t* d_keys, size_t n, uint32_t miss, uint32_t* d_out, void* stream) {
    if (!ix || (n && (!d_keys || !d_out))) return fail(UMGAP_ERR_INVALID, "umgap_index_lookup9_dev: null argument");
    if (ix->bound_to) return fail(UMGAP_ERR_INVALID, "the index is bound to a taxonomy (umgap_index_bind_taxonomy): its values are interval codes, not taxon ids; unbind it first");
    if (ix->kind != 9 && ix->kind != 10) return fail(UMGAP_ERR_INVALID, "umgap_index_lookup9_dev: not a 9-mer index");
    int rc = use_device(ix->device);
    if (rc) return rc;
    if (n) hipLaunchKernelGGL(k_lookup9, dim3(grid_for(n, 256, 256 * 8)), dim3(256), 0, (hipStream_t)stream, d_keys, n, ix->kind == 10 ? UMGAP_NONE : miss, d_out, ix->view);
    UMGAP_HIP(hipGetLastError());
    if (ix->kind == 10) return walk_map_positions(ix, d_keys, d_out, n, miss, (hipStream_t)stream);
    return UMGAP_OK;
}

// measurement-only entry: the lane-per-key variant
int umgap_index_lookup9_lane_dev(umgap_index* ix, const uint64_t* d_keys, size_t n, uint32_t miss, uint32_t* d_out, void* stream) {
    if (!ix || (n && (!d_keys || !d_out))) return fail(UMGAP_ERR_INVALID, "umgap_index_lookup9_lane_dev: null argument");
    if (ix->kind != 9) return fail(UMGAP_ERR_INVALID, "umgap_index_lookup9_lane_dev: not a plain 9-mer index");
    if (ix->bound_to) return fail(UMGAP_ERR_INVALID, "the index is bound to a taxonomy (umgap_index_bind_taxonomy): its values are interval codes, not taxon ids; unbind it first");
    int rc = use_device(ix->device);
    if (rc) return rc;
    if (n) hipLaunchKernelGGL(k_lookup9_lane, dim3(grid_for(n, 256, 256 * 8)), dim3(256), 0, (hipStream_t)stream, d_keys, n, miss, d_out, ix->view);
    UMGAP_HIP(hipGetLastError());
    return UMGAP_OK;
}

int umgap_index_lookup_prot9_dev(umgap_index* ix, const uint8_t* d_bytes, const uint64_t* d_offsets, const uint64_t* d_out_offsets,
                                 size_t n_prot, uint32_t miss, uint32_t* d_out, void* stream) {
    if (!ix || (n_prot && (!d_bytes || !d_offsets || !d_out_offsets || !d_out))) return fail(UMGAP_ERR_INVALID, "umgap_index_lookup_prot9_dev: null argument");
    if (ix->bound_to) return fail(UMGAP_ERR_INVALID, "the index is bound to a taxonomy (umgap_index_bind_taxonomy): its values are interval codes, not taxon ids; unbind it first");
    int rc = use_device(ix->device);
    if (rc) return rc;
    if (!n_prot) return UMGAP_OK;
    uint64_t n_windows = 0;
    UMGAP_HIP(hipMemcpyAsync(&n_windows, d_out_offsets + n_prot, sizeof(uint64_t), hipMemcpyDeviceToHost, (hipStream_t)stream));
    UMGAP_HIP(hipStreamSynchronize((hipStream_t)stream));
    if (ix->kind != 9 && ix->kind != 10) return fail(UMGAP_ERR_INVALID, "umgap_index_lookup_prot9_dev: not a 9-mer index");
    if (ix->kind == 10) return walk_lookup_prot9(ix, d_bytes, d_offsets, d_out_offsets, n_prot, (size_t)n_windows, miss, d_out, (hipStream_t)stream);   // walking (index_walk.hip)
    if (n_windows) hipLaunchKernelGGL(k_lookup_prot9, dim3(grid_for(n_windows)), dim3(256), 0, (hipStream_t)stream, d_bytes, d_offsets, d_out_offsets, n_prot, (size_t)n_windows, miss, d_out, ix->view);
    UMGAP_HIP(hipGetLastError());
    return UMGAP_OK;
}

int umgap_index_lookup9(umgap_index* ix, const uint64_t* keys, size_t n, uint32_t miss, uint32_t* out) {
    if (!ix || (n && (!keys || !out))) return fail(UMGAP_ERR_INVALID, "umgap_index_lookup9: null argument");
    if (ix->bound_to) return fail(UMGAP_ERR_INVALID, "the index is bound to a taxonomy (umgap_index_bind_taxonomy): its values are interval codes, not taxon ids; unbind it first");
    if (ix->kind != 9 && ix->kind != 10) return fail(UMGAP_ERR_INVALID, "umgap_index_lookup9: not a 9-mer index");
    const bool walk = ix->kind == 10;   // the table holds window numbers: the taxa come from the walk array in a second pass
    int rc = use_device(ix->device);
    if (rc) return rc;
    rc = ensure_stage(ix);
    if (rc) return rc;
    // An UPLOAD lane and a PROBE lane tied by events per staging slot (three slots): the keys of chunks c + 1 and c + 2 cross the
    // link while chunk c is probed and its answers go back.  Arrays in pinned memory (umgap_host_alloc, hipHostMalloc, a registered
    // range) are copied from / to directly; pageable ones go through the handle's pinned staging buffers with a host memcpy.
    const size_t cap = ix->stage_cap;
    constexpr int NS = 3;
    auto pinned = [](const void* p) {
        hipPointerAttribute_t a{};
        if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
        return a.type == hipMemoryTypeHost;
    };
    const bool in_pinned = n && pinned(keys) && pinned(keys + n - 1), out_pinned = n && pinned(out) && pinned(out + n - 1);
    struct Events {
        hipEvent_t uploaded[NS] = {}, done[NS] = {};
        ~Events() { for (hipEvent_t x : uploaded) if (x) (void)hipEventDestroy(x); for (hipEvent_t x : done) if (x) (void)hipEventDestroy(x); }
    } ev;
    for (int i = 0; i < NS; i++) {
        UMGAP_HIP(hipEventCreateWithFlags(&ev.uploaded[i], hipEventDisableTiming));
        UMGAP_HIP(hipEventCreateWithFlags(&ev.done[i], hipEventDisableTiming));
    }
    const hipStream_t up = ix->up_stream, run = ix->run_stream;
    size_t pending_off[NS] = {}, pending_len[NS] = {};
    auto drain = [&](int s) -> int {          // the answers of the slot's last chunk: waited for, and copied out when they went through staging
        if (!pending_len[s]) return UMGAP_OK;
        UMGAP_HIP(hipEventSynchronize(ev.done[s]));
        if (!out_pinned) par_memcpy(out + pending_off[s], (char*)ix->h_stage[s] + cap * 8, pending_len[s] * 4);
        pending_len[s] = 0;
        return UMGAP_OK;
    };
    size_t c = 0;
    for (size_t off = 0; off < n; off += cap, c++) {
        const int slot = (int)(c % NS);
        const size_t len = std::min(cap, n - off);
        rc = drain(slot);                     // the slot's previous tenant is through (upload included: done follows uploaded)
        if (rc) break;
        uint64_t* dk = (uint64_t*)ix->d_stage[slot];
        uint32_t* dout = (uint32_t*)((char*)ix->d_stage[slot] + cap * 8);
        const uint64_t* src = keys + off;
        if (!in_pinned) { par_memcpy(ix->h_stage[slot], keys + off, len * 8); src = (const uint64_t*)ix->h_stage[slot]; }
        UMGAP_HIP_BREAK(hipMemcpyAsync(dk, src, len * 8, hipMemcpyHostToDevice, up));
        UMGAP_HIP_BREAK(hipEventRecord(ev.uploaded[slot], up));
        UMGAP_HIP_BREAK(hipStreamWaitEvent(run, ev.uploaded[slot], 0));
        hipLaunchKernelGGL(k_lookup9, dim3(grid_for(len)), dim3(256), 0, run, dk, len, walk ? UMGAP_NONE : miss, dout, ix->view);
        if (walk && (rc = walk_map_positions(ix, dk, dout, len, miss, run))) break;
        UMGAP_HIP_BREAK(hipMemcpyAsync(out_pinned ? (void*)(out + off) : (void*)((char*)ix->h_stage[slot] + cap * 8), dout, len * 4, hipMemcpyDeviceToHost, run));
        UMGAP_HIP_BREAK(hipEventRecord(ev.done[slot], run));
        pending_off[slot] = off;
        pending_len[slot] = len;
    }
    for (int i = 0; i < NS; i++) { const int r2 = drain(i); if (!rc) rc = r2; }
    if (rc) { (void)hipStreamSynchronize(up); (void)hipStreamSynchronize(run); }
    return rc;
}

// ---- dump (printindex): every (key, taxon) of the table back on the host ------------------------
namespace {
inline uint64_t unmix45(uint64_t h) {    // inverse of mix45 (common.hpp)
    h ^= h >> 24;
    h = (h * inv_odd45(MIX_C2)) & MASK45;
    h ^= (h >> 21) ^ (h >> 42);
    h = (h * inv_odd45(MIX_C1)) & MASK45;
    h ^= h >> 23;
    return h;
}
}  // namespace

int umgap_index_dump9(const umgap_index* ix, uint64_t* keys45, uint32_t* taxa, size_t cap, size_t* n_out) {
    if (!ix || (ix->kind != 9 && ix->kind != 10) || !n_out) return fail(UMGAP_ERR_INVALID, "umgap_index_dump9: not a 9-mer index");
    if (ix->bound_to) return fail(UMGAP_ERR_INVALID, "the index is bound to a taxonomy (umgap_index_bind_taxonomy): its values are interval codes, not taxon ids; unbind it first");
    if (ix->view.home_lo != 0 || ix->view.home_hi != ix->view.nb) return fail(UMGAP_ERR_INVALID, "umgap_index_dump9: handle holds a shard");
    int rc = use_device(ix->device);
    if (rc) return rc;
    const uint32_t m = ix->view.m, nb = ix->view.nb;
    size_t n = 0;
    const size_t chunk_buckets = 1u << 20;
    std::vector<uint64_t> buf(chunk_buckets * SLOTS), ck;
    std::vector<uint32_t> cv;
    // kind 10: the table's values are entry numbers; a chunk's keys go through the walk array on the device.  An index whose entries were
    // valued after the fact (umgap_walk_builder_new_deferred) also holds the proteome's 9-mers that are no keys: value 0, left out here.
    uint32_t* d_tmp = nullptr;
    uint64_t* d_tk = nullptr;
    auto done = [&](int code) { if (d_tmp) (void)hipFree(d_tmp); if (d_tk) (void)hipFree(d_tk); return code; };
    const size_t step = chunk_buckets * SLOTS;
    if (ix->kind == 10 && (hipMalloc((void**)&d_tmp, step * 4) != hipSuccess || hipMalloc((void**)&d_tk, step * 8) != hipSuccess))
        return done(fail(UMGAP_ERR_NOMEM, "umgap_index_dump9: hipMalloc failed"));
    auto flush = [&]() -> int {        // ck / cv: a chunk's keys and table values -> the output
        if (ix->kind == 10 && !ck.empty()) {
            if (hipMemcpyAsync(d_tmp, cv.data(), ck.size() * 4, hipMemcpyHostToDevice, ix->stream) != hipSuccess ||
                hipMemcpyAsync(d_tk, ck.data(), ck.size() * 8, hipMemcpyHostToDevice, ix->stream) != hipSuccess) return fail(UMGAP_ERR_DEVICE, "umgap_index_dump9: copy failed");
            int r = walk_map_positions(ix, d_tk, d_tmp, ck.size(), 0u, ix->stream);
            if (r) return r;
            if (hipMemcpyAsync(cv.data(), d_tmp, ck.size() * 4, hipMemcpyDeviceToHost, ix->stream) != hipSuccess || hipStreamSynchronize(ix->stream) != hipSuccess)
                return fail(UMGAP_ERR_DEVICE, "umgap_index_dump9: copy failed");
        }
        for (size_t i = 0; i < ck.size(); i++) {
            if (ix->walk_zero_absent && cv[i] == 0u) continue;
            if (n < cap) { keys45[n] = ck[i]; taxa[n] = cv[i]; }
            n++;
        }
        ck.clear();
        cv.clear();
        return UMGAP_OK;
    };
    for (size_t b0 = 0; b0 < nb; b0 += chunk_buckets) {
        size_t cnt = std::min<size_t>(chunk_buckets, nb - b0);
        if (hipMemcpy(buf.data(), ix->d_slots + b0 * SLOTS, cnt * 64, hipMemcpyDeviceToHost) != hipSuccess) return done(fail(UMGAP_ERR_DEVICE, "umgap_index_dump9: copy failed"));
        for (size_t i = 0; i < cnt * SLOTS; i++) {
            uint64_t e = buf[i];
            if (e == EMPTY_SLOT) continue;
            uint32_t lo = (uint32_t)e, d = (lo >> R_BITS) & 7u, r = lo & R_MASK;
            uint64_t bucket = b0 + i / SLOTS, home = (bucket + nb - d) % nb;
            uint64_t h = ((home / m) << R_BITS) | r;
            ck.push_back(unmix45(h));
            cv.push_back((uint32_t)(e >> 32));
        }
        if ((rc = flush())) return done(rc);
    }
    if (ix->n_stash) {
        std::vector<uint64_t> sk(ix->n_stash);
        std::vector<uint32_t> sv(ix->n_stash);
        if (hipMemcpy(sk.data(), ix->d_stash_keys, sk.size() * 8, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(sv.data(), ix->d_stash_vals, sv.size() * 4, hipMemcpyDeviceToHost) != hipSuccess)
            return done(fail(UMGAP_ERR_DEVICE, "umgap_index_dump9: copy failed"));
        for (size_t i0 = 0; i0 < sk.size(); i0 += step) {
            for (size_t i = i0; i < std::min(sk.size(), i0 + step); i++) {
                uint64_t home = sk[i] >> R_BITS, r = sk[i] & R_MASK;
                ck.push_back(unmix45(((home / m) << R_BITS) | r));
                cv.push_back(sv[i]);
            }
            if ((rc = flush())) return done(rc);
        }
    }
    *n_out = n;
    return done(UMGAP_OK);
}

// EXPERIMENT kept beside the product (tools/gather_streams.py; not in the public header): n random 64-byte quad probes with
// read_bytes read from d_side_in and write_bytes written to d_side_out per probe (0, 4, 8 or 16; the buffers hold side_bytes each:
// n * bytes and a little more -- the streams stop where the buffers end)
int umgap_bench_gather_rw_dev(umgap_index* ix, size_t n, uint32_t read_bytes, uint32_t write_bytes, const void* d_side_in, void* d_side_out, size_t side_bytes,
                              uint32_t* d_out, void* stream) {
    if (!ix || !d_out || (read_bytes && !d_side_in) || (write_bytes && !d_side_out)) return fail(UMGAP_ERR_INVALID, "umgap_bench_gather_rw_dev: null argument");
    int rc = use_device(ix->device);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    const uint32_t nb = ix->view.n_resident;
    const int g = 256 * 8;
    const u32x4* in = (const u32x4*)d_side_in;
    u32x4* outp = (u32x4*)d_side_out;
#define UMGAP_RW(SR, SW) if (read_bytes == SR && write_bytes == SW) { hipLaunchKernelGGL((k_gather_rw<SR, SW>), dim3(g), dim3(256), 0, s, ix->view.slots, nb, n, in, outp, side_bytes / 16, d_out); UMGAP_HIP(hipGetLastError()); return UMGAP_OK; }
    if (read_bytes == 0 && write_bytes == 0x204) {      // 0x200 + 4: the 4-byte stream as no-return 8-byte atomic exchanges
        hipLaunchKernelGGL(k_gather_w_atomic, dim3(g), dim3(256), 0, s, ix->view.slots, nb, n, (unsigned long long*)d_side_out, side_bytes / 8, d_out);
        UMGAP_HIP(hipGetLastError());
        return UMGAP_OK;
    }
    if (read_bytes == 0 && (write_bytes >> 8) == 3) {      // 0x300 + mode: round 5's variants of the 4-byte write stream (k_gather_w_variant)
        const uint32_t mode = write_bytes & 0xFFu;
        const size_t vecs = side_bytes / 16;
#define UMGAP_WV(M) if (mode == M) { hipLaunchKernelGGL((k_gather_w_variant<M>), dim3(g), dim3(256), 0, s, ix->view.slots, nb, n, outp, vecs, d_out); UMGAP_HIP(hipGetLastError()); return UMGAP_OK; }
        UMGAP_WV(1) UMGAP_WV(2) UMGAP_WV(3) UMGAP_WV(4) UMGAP_WV(8) UMGAP_WV(17) UMGAP_WV(18) UMGAP_WV(19)
#undef UMGAP_WV
        return fail(UMGAP_ERR_INVALID, "umgap_bench_gather_rw_dev: unknown write-stream variant");
    }
    if (read_bytes == 0 && (write_bytes == 0x104 || write_bytes == 0x110)) {      // 0x100 + bytes: the deferred-store form
        const size_t vecs = side_bytes / 16 - 64;                                  // (64 vectors at the end: where idle lanes' stores go)
        if (write_bytes == 0x104) hipLaunchKernelGGL((k_gather_w_deferred<4>), dim3(g), dim3(256), 0, s, ix->view.slots, nb, n, outp, vecs, d_out);
        else hipLaunchKernelGGL((k_gather_w_deferred<16>), dim3(g), dim3(256), 0, s, ix->view.slots, nb, n, outp, vecs, d_out);
        UMGAP_HIP(hipGetLastError());
        return UMGAP_OK;
    }
    UMGAP_RW(0, 0) UMGAP_RW(4, 0) UMGAP_RW(8, 0) UMGAP_RW(16, 0) UMGAP_RW(0, 4) UMGAP_RW(0, 8) UMGAP_RW(0, 16) UMGAP_RW(8, 4) UMGAP_RW(8, 8) UMGAP_RW(16, 16)
#undef UMGAP_RW
    return fail(UMGAP_ERR_INVALID, "umgap_bench_gather_rw_dev: unsupported combination");
}

int umgap_bench_gather_dev(umgap_index* ix, size_t n, uint32_t bytes_per_probe, uint32_t* d_out, void* stream) {
    if (!ix || !d_out) return fail(UMGAP_ERR_INVALID, "umgap_bench_gather_dev: null argument");
    int rc = use_device(ix->device);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    // bits 24-27: footprint shift -- the probes are confined to the first n_resident >> shift buckets (TLB / cache reach experiments)
    uint32_t nb = ix->view.n_resident >> ((bytes_per_probe >> 24) & 0xFu);
    if (nb < 16) nb = 16;
    bytes_per_probe &= ~(0xFu << 24);
    int g = 256 * 8;
    if (bytes_per_probe >> 28) {   // bits 28+: 1 + cache-policy variant of the 64-byte, 8-in-flight gather (k_gather_policy)
        switch ((bytes_per_probe >> 28) - 1) {
            case 0: hipLaunchKernelGGL(k_gather_policy<0>, dim3(g), dim3(256), 0, s, ix->view.slots, nb, n, d_out); break;
            case 1: hipLaunchKernelGGL(k_gather_policy<1>, dim3(g), dim3(256), 0, s, ix->view.slots, nb, n, d_out); break;
            case 2: hipLaunchKernelGGL(k_gather_policy<2>, dim3(g), dim3(256), 0, s, ix->view.slots, nb, n, d_out); break;
            case 3: hipLaunchKernelGGL(k_gather_policy<3>, dim3(g), dim3(256), 0, s, ix->view.slots, nb, n, d_out); break;
            case 4: hipLaunchKernelGGL(k_gather_policy<4>, dim3(g), dim3(256), 0, s, ix->view.slots, nb, n, d_out); break;
            case 5: hipLaunchKernelGGL(k_gather_policy<5>, dim3(g), dim3(256), 0, s, ix->view.slots, nb, n, d_out); break;
            case 6: hipLaunchKernelGGL(k_gather_policy<6>, dim3(g), dim3(256), 0, s, ix->view.slots, nb, n, d_out); break;
            default: hipLaunchKernelGGL(k_gather_policy<7>, dim3(g), dim3(256), 0, s, ix->view.slots, nb, n, d_out); break;
        }
        UMGAP_HIP(hipGetLastError());
        return UMGAP_OK;
    }
    if (bytes_per_probe >> 16) {   // bits 16+: that many probes in flight per lane, no store per probe (pure random-read rate)
        const uint32_t fl = (bytes_per_probe >> 16) & 0xFFu, bytes = bytes_per_probe & 0xFFFFu;
        if (bytes == 65 || bytes == 66 || bytes == 68 || bytes == 72) {      // 64 + KS: the gather with KS scalar-path probes per wave step beside it (round 5)
            if (bytes == 65) hipLaunchKernelGGL((k_gather_scalar_mix<1>), dim3(g), dim3(256), 0, s, ix->view.slots, nb, n, d_out);
            else if (bytes == 66) hipLaunchKernelGGL((k_gather_scalar_mix<2>), dim3(g), dim3(256), 0, s, ix->view.slots, nb, n, d_out);
            else if (bytes == 68) hipLaunchKernelGGL((k_gather_scalar_mix<4>), dim3(g), dim3(256), 0, s, ix->view.slots, nb, n, d_out);
            else hipLaunchKernelGGL((k_gather_scalar_mix<0>), dim3(g), dim3(256), 0, s, ix->view.slots, nb, n, d_out);
            UMGAP_HIP(hipGetLastError());
            return UMGAP_OK;
        }
        if (bytes == 64 && fl == 2) hipLaunchKernelGGL((k_gather_mlp<4, 2>), dim3(g), dim3(256), 0, s, ix->view.slots, nb, n, d_out);
        else if (bytes == 64 && fl == 4) hipLaunchKernelGGL((k_gather_mlp<4, 4>), dim3(g), dim3(256), 0, s, ix->view.slots, nb, n, d_out);
        else if (bytes == 64 && fl == 8) hipLaunchKernelGGL((k_gather_mlp<4, 8>), dim3(g), dim3(256), 0, s, ix->view.slots, nb, n, d_out);
        else if (bytes == 128 && fl == 4) hipLaunchKernelGGL((k_gather_mlp<8, 4>), dim3(g), dim3(256), 0, s, ix->view.slots, nb - 1, n, d_out);
        else if (bytes == 32 && fl == 8) hipLaunchKernelGGL((k_gather_mlp<2, 8>), dim3(g), dim3(256), 0, s, ix->view.slots, nb, n, d_out);
        else return fail(UMGAP_ERR_INVALID, "umgap_bench_gather_dev: unsupported bytes / in-flight combination");
        UMGAP_HIP(hipGetLastError());
        return UMGAP_OK;
    }
    switch (bytes_per_probe) {
        case 16: hipLaunchKernelGGL(k_gather<1>, dim3(g), dim3(256), 0, s, ix->view.slots, nb, n, d_out); break;
        case 32: hipLaunchKernelGGL(k_gather<2>, dim3(g), dim3(256), 0, s, ix->view.slots, nb, n, d_out); break;
        case 64: hipLaunchKernelGGL(k_gather<4>, dim3(g), dim3(256), 0, s, ix->view.slots, nb, n, d_out); break;
        case 128: hipLaunchKernelGGL(k_gather<8>, dim3(g), dim3(256), 0, s, ix->view.slots, nb, n, d_out); break;
        case 256: hipLaunchKernelGGL(k_gather<16>, dim3(g), dim3(256), 0, s, ix->view.slots, nb, n, d_out); break;
        default: return fail(UMGAP_ERR_INVALID, "umgap_bench_gather_dev: bytes_per_probe must be 16, 32, 64, 128 or 256");
    }
    UMGAP_HIP(hipGetLastError());
    return UMGAP_OK;
}

}  // extern "C"
