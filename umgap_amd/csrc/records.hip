// records.hip — the 9-mer pipeline over a block of FASTA records held in host memory:
//   prot2kmer2lca -o <index> | seedextend -s S -g G | uniq [-d D] | taxa2agg <options> <taxons>
// (scripts/umgap-analyse.sh:257-310) as ONE C-ABI call.  The text between the four reference
// commands (84 decimal taxon ids per read pair, three pipes) never exists: the host hands over the
// protein fragments (what prot2kmer2lca reads, src/commands/prot2kmer2lca.rs:172-179) and which
// consecutive fragments `uniq` would merge (src/commands/uniq.rs:66-79); the per-window hits stay in
// HBM between the probe, seedextend (src/commands/seedextend.rs:99-148, in place) and the per-read
// aggregate (src/commands/taxa2agg.rs:160-180).  Only one taxon + status per read travels back.
//
// Shape of the call (round 2): nothing is rebuilt on the host.  The block is cut into chunks of whole reads
// (~32 MB of residues); chunks c+1 and c+2 are uploaded straight from the caller's arrays (pin them with umgap_host_alloc
// for full PCIe rate) on an upload stream while chunk c runs on the compute stream (events per staging slot).  On the device the hit of the window that
// STARTS at byte b lives at hits[b] -- fragment f owns hits[off[f] .. off[f+1]), its last 8 positions (no window
// starts there) stay 0, which taxa2agg drops (src/commands/taxa2agg.rs:169) -- so no window prefix sums are needed:
// (Round 3: a call whose reads all have the same number of fragments and fit the fused surface-B kernel takes ONE launch per chunk
// instead -- k_records_fit + k_classify_prot9, further down; what follows is the general route.)
//   k_frag_prepare   fragment -> {start, length} (validated) + the tail positions marked in a bitmap
//   k_lookup_bytes9  every unmarked byte position: 9 residues packed from LDS -> quad probe -> hits[b]
//   k_seedextend_frags / k_read_offsets / the aggregate kernels / k_mark_seed_panics
#include "bound.hpp"
#include "host_copy.hpp"
#include "probe_common.hpp"
#include "prot_limits.hpp"
#include "seedextend.hpp"
#include "taxonomy.hpp"
#include "walk_ref.hpp"

#include <algorithm>
#include <vector>

namespace umgap {
int ensure_stage(umgap_index* ix);   // index.hip
int mark_seedextend_panics(const uint32_t* d_kept, const uint64_t* d_read_frags, size_t n_reads, uint32_t* d_out_taxon, uint32_t* d_out_status,
                           hipStream_t stream);
}

using namespace umgap;
using namespace umgap::probe;

namespace {

// fragment f of the chunk (global fragment f0 + f): start/length relative to the chunk's first byte, checked against the
// chunk (a decreasing or overrunning offset raises *bad and empties the fragment); its last min(len, 8) byte positions
// start no window: marked in `tails` (bit b of word b / 32).
__global__ __launch_bounds__(256) void k_frag_prepare(const uint64_t* __restrict__ off, size_t n_frag, uint64_t base, uint32_t chunk_bytes,
                                                      uint32_t* __restrict__ fstart, uint32_t* __restrict__ flen, uint32_t* __restrict__ tails,
                                                      uint32_t* __restrict__ bad) {
    size_t f = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (; f < n_frag; f += step) {
        const uint64_t a = off[f], b = off[f + 1];
        uint32_t s = 0, l = 0;
        if (a < base || b < a || b - base > chunk_bytes) atomicExch(bad, 1u);
        else { s = (uint32_t)(a - base); l = (uint32_t)(b - a); }
        fstart[f] = s;
        flen[f] = l;
        const uint32_t t0 = l > 8 ? s + l - 8 : s, t1 = s + l;    // [t0, t1): at most 8 bits, at most two words
        if (t1 > t0) {
            const uint32_t w0 = t0 >> 5, w1 = (t1 - 1) >> 5;
            const uint32_t lo = 0xFFFFFFFFu << (t0 & 31), hi = 0xFFFFFFFFu >> (31 - ((t1 - 1) & 31));
            if (w0 == w1) atomicOr(&tails[w0], lo & hi);
            else { atomicOr(&tails[w0], lo); atomicOr(&tails[w1], hi); }
        }
    }
}

// hits[b] = value of the 9-mer bytes[b .. b+9) (0 = miss; windows holding a byte outside 'A'..'Z' miss,
// src/commands/prot2kmer2lca.rs:172-179 + the packing convention of include/umgap_hip.h) for every position that starts
// a window, 0 elsewhere.  A wave takes 128 consecutive positions per step: their 136 bytes are staged in the wave's LDS
// strip, two probe batches of 64 go out back to back (8 sectors in flight per quad), results land coalesced.
// n_avail >= n_bytes: bytes that may be read (a chunk cut inside a fragment brings the 8 bytes behind its last position along).
// W.E != nullptr: a walk-order index -- the table's answer is the number of the key's entry (walk_ref.hpp) and the taxon comes from there.
__global__ __launch_bounds__(256) void k_lookup_bytes9(TableView T, const uint8_t* __restrict__ bytes, const uint32_t* __restrict__ tails,
                                                       uint32_t n_bytes, uint32_t n_avail, uint32_t* __restrict__ hits, uint32_t miss, WalkRef W) {
    __shared__ uint8_t lds[4][144];
    const uint32_t wv = threadIdx.x >> 6, lane = threadIdx.x & 63, q = lane >> 2, j = lane & 3;
    uint8_t* wb = lds[wv];
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t p0 = wave * 128; p0 < n_bytes; p0 += nwaves * 128) {
        // stage bytes [p0, p0 + 136)
        for (uint32_t i = lane; i < 136; i += 64) wb[i] = p0 + i < n_avail ? bytes[p0 + i] : (uint8_t)0;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const uint32_t tw0 = tails[(p0 >> 5) + (lane >> 5)], tw1 = p0 + 64 < n_bytes ? tails[((p0 + 64) >> 5) + (lane >> 5)] : 0xFFFFFFFFu;
        ProbeBatch B[2];
        uint32_t res[2][4];
        uint64_t wkey[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const uint32_t p = p0 + 64 * h + lane;
            const bool starts = p < n_bytes && !(((h ? tw1 : tw0) >> (lane & 31)) & 1u);
            uint64_t key = 0;
            bool ok = starts;
#pragma unroll
            for (int c = 0; c < 9; c++) {
                const uint32_t code = (uint32_t)wb[64 * h + lane + c] - 'A';
                ok = ok && code < 26u;
                key = (key << 5) | (code & 31u);
            }
            uint32_t local = 0, tag = 0, valid = 0;
            if (ok) valid = locate(T, key, local, tag) ? 1u : 0u;
            wkey[h] = key;
            probe_issue(T, local, tag | (valid << 31), lane, B[h]);
        }
        probe_eval_multi<2>(T, B, W.E ? UMGAP_NONE : miss, lane, res);
        if (W.E) {      // the answer to lane L's key sits, quad-uniform, in quad L & 15 as round L >> 4: back to its lane, then through the walk array
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int src = 4 * (int)(lane & 15u);
                const uint32_t v0 = (uint32_t)__shfl((int)res[h][0], src, 64), v1 = (uint32_t)__shfl((int)res[h][1], src, 64),
                               v2 = (uint32_t)__shfl((int)res[h][2], src, 64), v3 = (uint32_t)__shfl((int)res[h][3], src, 64);
                const uint32_t r4 = lane >> 4, q32 = r4 == 0 ? v0 : (r4 == 1 ? v1 : (r4 == 2 ? v2 : v3));
                const uint32_t p = p0 + 64 * h + lane;
                if (p < n_bytes) hits[p] = walk_taxon(W, q32, wkey[h], miss);
            }
        } else if (j == 0) {
#pragma unroll
            for (int h = 0; h < 2; h++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const uint32_t p = p0 + 64 * h + 16 * r + q;
                    if (p < n_bytes) hits[p] = res[h][r];
                }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// seedextend per fragment over its windows hits[start .. start + len - 8) (one lane per fragment, as k_seedextend)
__global__ __launch_bounds__(256) void k_seedextend_frags(const uint32_t* __restrict__ fstart, const uint32_t* __restrict__ flen, size_t n_frag,
                                                          uint32_t min_seed, uint32_t max_gap, uint32_t* __restrict__ hits, uint32_t* __restrict__ kept) {
    size_t f = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (; f < n_frag; f += step) {
        const uint32_t l = flen[f];
        // a fragment without windows is an empty record: the state machine over [sentinel] emits nothing (and cannot panic)
        const uint32_t k = l > 8 ? seedextend_record(hits + fstart[f], l - 8, min_seed, max_gap) : 0u;
        if (kept) kept[f] = k;
    }
}

// read r of the chunk owns hits[roff[r] .. roff[r+1]); rf = the chunk's slice of read_frags (n_reads + 1 entries, global
// fragment numbers), checked to be non-decreasing and inside the chunk's fragments [f0, f0 + n_frag]
__global__ __launch_bounds__(256) void k_read_offsets(const uint64_t* __restrict__ rf, size_t n_reads, uint64_t f0, size_t n_frag,
                                                      const uint32_t* __restrict__ fstart, uint32_t chunk_bytes, uint64_t* __restrict__ roff,
                                                      uint64_t* __restrict__ rfrag_local, uint32_t* __restrict__ bad) {
    size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (; r <= n_reads; r += step) {
        const uint64_t g = rf[r];
        uint64_t f = 0;
        if (g < f0 || g - f0 > n_frag || (r && g < rf[r - 1])) atomicExch(bad, 1u);
        else f = g - f0;
        roff[r] = f == n_frag ? (uint64_t)chunk_bytes : (uint64_t)fstart[f];
        rfrag_local[r] = f;
    }
}

// The FUSED route of the record path: when every read of the call has the same number of fragments and fits the fused surface-B
// kernel (prot_limits.hpp), a chunk is ONE launch of k_classify_prot9 (classify.hip: bytes staged in LDS, rolling keys, probe,
// seedextend and aggregate in one kernel; over a walk-order index the walking variant) instead of the six kernels above with a hit
// per byte position in HBM between them.  This kernel checks what the host did not look at -- offsets non-decreasing and inside the
// chunk (flags[0], as k_frag_prepare / k_read_offsets), every read of exactly F fragments within the limits of the short variant of
// the fused kernel (flags[1]) and of its LONG variant (flags[2]; prot_limits.hpp) -- and writes the fragment offsets relative to the
// chunk's first byte.  BOTH variants are then launched, gated on the flags: the device picks the one that fits, no host round trip.
__global__ __launch_bounds__(256) void k_records_fit(const uint64_t* __restrict__ off, size_t n_frag, const uint64_t* __restrict__ rf, size_t n_reads, uint64_t f0,
                                                     uint32_t F, uint64_t base, uint32_t chunk_bytes, uint64_t* __restrict__ rel, uint32_t* __restrict__ flags) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (size_t f = i; f <= n_frag; f += step) {
        const uint64_t a = off[f];
        const bool ok = a >= base && a - base <= chunk_bytes && (f == 0 || off[f - 1] <= a);
        if (!ok) atomicExch(flags, 1u);
        rel[f] = ok ? a - base : 0;
    }
    for (size_t r = i; r < n_reads; r += step) {
        const uint64_t ga = rf[r], gb = rf[r + 1];
        if (ga < f0 || gb < ga || gb - f0 > n_frag) { atomicExch(flags, 1u); continue; }
        if (gb - ga != F) { atomicExch(flags + 1, 1u); atomicExch(flags + 2, 1u); continue; }
        const uint64_t fa = ga - f0;
        uint64_t windows = 0;
        bool sane = true;
        for (uint32_t t = 0; t < F; t++) {
            const uint64_t a = off[fa + t], b = off[fa + t + 1];
            sane = sane && b >= a;
            if (b >= a + 9) windows += b - a - 8;
        }
        const uint64_t a0 = off[fa], b0 = off[fa + F];
        if (!sane || b0 < a0) { atomicExch(flags + 1, 1u); atomicExch(flags + 2, 1u); continue; }
        if (b0 - a0 > PROT_MAX_BYTES || windows > PROT_MAX_WINDOWS || F > PROT_MAX_FRAGS) atomicExch(flags + 1, 1u);       // not for the short variant
        if (b0 - a0 > PROT_LONG_BYTES || windows > PROT_LONG_WINDOWS || F > PROT_LONG_FRAGS) atomicExch(flags + 2, 1u);    // nor for the LONG one
    }
}
// a read the fused kernel could not take after all (hybrid / MRTL: more distinct taxa than its scratch holds) sends the call
// through the general kernels
__global__ __launch_bounds__(256) void k_records_any_unsupported(const uint32_t* __restrict__ status, size_t n, uint32_t* __restrict__ flags) {
    if (flags[0] | flags[2]) return;       // (no fused launch ran: the statuses are stale)
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    bool any = false;
    for (; i < n; i += step) any = any || status[i] == UMGAP_READ_UNSUPPORTED;
    if (any) atomicExch(flags + 3, 1u);    // word 3: the fused launch ran and left a read to the general kernels
}

// umgap_index_lookup_records: the fragments [off[f], off[f + 1]) that touch the byte range [c0, c1) of the call; the last min(len, 8)
// positions of each start no window: marked in `tails` (bit p - c0)
__global__ __launch_bounds__(256) void k_mark_tails(const uint64_t* __restrict__ off, size_t n_frag, uint64_t c0, uint64_t c1, uint32_t* __restrict__ tails) {
    size_t f = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (; f < n_frag; f += step) {
        const uint64_t a = off[f], b = off[f + 1];
        if (b <= a) continue;
        uint64_t t0 = b - a > 8 ? b - 8 : a, t1 = b;
        if (t0 < c0) t0 = c0;
        if (t1 > c1) t1 = c1;
        for (uint64_t p = t0; p < t1; p++) atomicOr(&tails[(p - c0) >> 5], 1u << ((p - c0) & 31));
    }
}

// the general route on a table bound to the taxonomy: the probe returned interval codes (bound.hpp; 0 = miss), the kernels behind it
// want taxon ids
__global__ __launch_bounds__(256) void k_codes_to_ids(uint32_t* __restrict__ hits, size_t n, const uint32_t* __restrict__ id_of_pre) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += step) {
        const uint32_t v = hits[i];
        if (v) hits[i] = id_of_pre[bound::pre_of(v)];
    }
}

struct Slot {
    uint8_t* bytes = nullptr;
    uint64_t *off = nullptr, *off_rel = nullptr, *rf = nullptr, *roff = nullptr, *rfl = nullptr;
    uint32_t *fstart = nullptr, *flen = nullptr, *tails = nullptr, *hits = nullptr, *kept = nullptr, *ot = nullptr, *os = nullptr, *bad = nullptr;
};

inline size_t pad256(size_t b) { return (b + 255) & ~(size_t)255; }

}  // namespace

// pinned host memory for the arrays handed to the host-pointer entry points (full-rate, truly asynchronous PCIe copies)
extern "C" int umgap_host_alloc(size_t bytes, void** out) {
    if (!out) return fail(UMGAP_ERR_INVALID, "umgap_host_alloc: null argument");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(UMGAP_ERR_DEVICE, "umgap_host_alloc: no usable GPU");
    hipError_t e = hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocPortable);
    if (e != hipSuccess) { *out = nullptr; return fail(UMGAP_ERR_NOMEM, std::string("hipHostMalloc: ") + hipGetErrorString(e)); }
    return UMGAP_OK;
}
extern "C" void umgap_host_free(void* p) {
    if (p) (void)hipHostFree(p);
}

// The windows of protein records, looked up where the bytes are: what `prot2kmer2lca` does per record (src/commands/prot2kmer2lca.rs:
// 172-179) for a whole block of records in one call.  The host hands over the residues (1.2 bytes per window instead of an 8-byte
// packed key per window, and no per-window work on the host at all); out_hits[b - frag_offsets[0]] = value of the 9-mer that starts
// at byte b, for every b that starts a window of its fragment (b + 9 <= end of the fragment); miss_value where the window is not
// [A-Z]{9} or not a key; positions that start no window hold miss_value too.  Chunks of ~3.5 MB of residues over the upload lane
// and the probe lane of the handle (three staging slots); a chunk may end inside a fragment (it brings 8 bytes of lookahead along).
extern "C" int umgap_index_lookup_records(umgap_index* ix, const uint8_t* bytes, const uint64_t* frag_offsets, size_t n_fragments, uint32_t miss_value,
                                          uint32_t* out_hits) {
    if (!ix || (n_fragments && (!frag_offsets || !out_hits))) return fail(UMGAP_ERR_INVALID, "umgap_index_lookup_records: null argument");
    if (ix->bound_to) return fail(UMGAP_ERR_INVALID, "the index is bound to a taxonomy (umgap_index_bind_taxonomy): its values are interval codes, not taxon ids; unbind it first");
    if (ix->kind != 9 && ix->kind != 10) return fail(UMGAP_ERR_INVALID, "umgap_index_lookup_records: not a 9-mer index");
    if (!n_fragments) return UMGAP_OK;
    const bool walk = ix->kind == 10;
    const uint64_t base = frag_offsets[0];
    for (size_t f = 0; f < n_fragments; f++)
        if (frag_offsets[f + 1] < frag_offsets[f]) return fail(UMGAP_ERR_INVALID, "umgap_index_lookup_records: frag_offsets must not decrease");
    const uint64_t total = frag_offsets[n_fragments] - base;
    if (!total) return UMGAP_OK;
    if (!bytes) return fail(UMGAP_ERR_INVALID, "umgap_index_lookup_records: null bytes");
    int rc = use_device(ix->device);
    if (rc) return rc;
    rc = ensure_stage(ix);
    if (rc) return rc;
    // a staging slot (24 MiB): [hits 14 MiB][bytes 3.5 MiB + lookahead][offsets 2 MiB][tails 0.5 MiB]
    constexpr int NS = 3;
    constexpr size_t CB = 3584u << 10, CF = (256u << 10) - 1, O_BYTES = CB * 4, O_OFF = O_BYTES + CB + 4096, O_TAILS = O_OFF + (CF + 1) * 8, SLOT = O_TAILS + CB / 8 + 4096;
    static_assert(SLOT <= (size_t)(2u << 20) * 12, "a chunk must fit a staging slot");
    auto pinned = [](const void* p) {
        hipPointerAttribute_t a{};
        if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
        return a.type == hipMemoryTypeHost;
    };
    const bool in_pinned = pinned(bytes + base) && pinned(bytes + base + total - 1) && pinned(frag_offsets) && pinned(frag_offsets + n_fragments);
    const bool out_pinned = pinned(out_hits) && pinned(out_hits + total - 1);
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
    auto drain = [&](int k) -> int {
        if (!pending_len[k]) return UMGAP_OK;
        UMGAP_HIP(hipEventSynchronize(ev.done[k]));
        if (!out_pinned) par_memcpy(out_hits + pending_at[k], ix->h_stage[k], pending_len[k] * 4);
        pending_len[k] = 0;
        return UMGAP_OK;
    };
    const TableView T = ix->view;
    size_t f_lo = 0;          // first fragment that ends behind the chunk's first byte
    size_t c = 0;
    for (uint64_t c0 = 0; c0 < total; c++) {
        const int slot = (int)(c % NS);
        while (f_lo < n_fragments && frag_offsets[f_lo + 1] - base <= c0) f_lo++;
        uint64_t c1 = std::min<uint64_t>(total, c0 + CB);
        // fragments that touch [c0, c1): [f_lo, f_hi)
        size_t f_hi = (size_t)(std::lower_bound(frag_offsets + f_lo, frag_offsets + n_fragments, base + c1) - frag_offsets);
        if (f_hi - f_lo > CF) { f_hi = f_lo + CF; c1 = frag_offsets[f_hi] - base; }      // many short fragments: the chunk ends with fragment f_hi - 1
        if (c1 <= c0) { c0 = c1 = frag_offsets[f_hi] - base; continue; }                   // (CF empty fragments in a row)
        const uint64_t nb = c1 - c0, navail = std::min<uint64_t>(total - c0, nb + 8);
        const size_t nf = f_hi - f_lo;
        rc = drain(slot);
        if (rc) break;
        char* D = (char*)ix->d_stage[slot];
        uint32_t* d_hits = (uint32_t*)D;
        uint8_t* d_bytes = (uint8_t*)(D + O_BYTES);
        uint64_t* d_off = (uint64_t*)(D + O_OFF);
        uint32_t* d_tails = (uint32_t*)(D + O_TAILS);
        const uint8_t* src_b = bytes + base + c0;
        const uint64_t* src_o = frag_offsets + f_lo;
        if (!in_pinned) {          // the pinned slot holds [bytes | offsets] on the way up (the answers of its last tenant have been drained)
            char* H = (char*)ix->h_stage[slot];
            memcpy(H, src_b, navail);
            memcpy(H + CB + 4096, src_o, (nf + 1) * 8);
            src_b = (const uint8_t*)H;
            src_o = (const uint64_t*)(H + CB + 4096);
        }
        UMGAP_HIP_BREAK(hipMemcpyAsync(d_bytes, src_b, navail, hipMemcpyHostToDevice, up));
        UMGAP_HIP_BREAK(hipMemcpyAsync(d_off, src_o, (nf + 1) * 8, hipMemcpyHostToDevice, up));
        UMGAP_HIP_BREAK(hipEventRecord(ev.uploaded[slot], up));
        UMGAP_HIP_BREAK(hipStreamWaitEvent(run, ev.uploaded[slot], 0));
        UMGAP_HIP_BREAK(hipMemsetAsync(d_tails, 0, (nb / 32 + 8) * 4, run));
        hipLaunchKernelGGL(k_mark_tails, dim3((unsigned)std::min<size_t>((nf + 255) / 256, 1024)), dim3(256), 0, run, (const uint64_t*)d_off, nf, base + c0, base + c1, d_tails);
        hipLaunchKernelGGL(k_lookup_bytes9, dim3((unsigned)std::min<uint64_t>((nb + 511) / 512, 2048)), dim3(256), 0, run, T, (const uint8_t*)d_bytes, (const uint32_t*)d_tails,
                           (uint32_t)nb, (uint32_t)navail, d_hits, miss_value, walk ? walk_ref_of(ix) : WalkRef{});
        UMGAP_HIP_BREAK(hipGetLastError());
        UMGAP_HIP_BREAK(hipMemcpyAsync(out_pinned ? (void*)(out_hits + c0) : ix->h_stage[slot], d_hits, nb * 4, hipMemcpyDeviceToHost, run));
        UMGAP_HIP_BREAK(hipEventRecord(ev.done[slot], run));
        pending_at[slot] = c0;
        pending_len[slot] = nb;
        c0 = c1;
    }
    for (int i = 0; i < NS; i++) { const int r2 = drain(i); if (!rc) rc = r2; }
    if (rc) { (void)hipStreamSynchronize(up); (void)hipStreamSynchronize(run); }
    return rc;
}

static thread_local int g_last_route = 0;

// The residues of a call: ONE array indexed by the fragment offsets (umgap_classify_records), or pieces laid end to end
// (umgap_classify_records_sg): piece i holds the bytes [ends[i - 1], ends[i]) of the concatenation (ends[-1] = 0).
struct ByteParts {
    const uint8_t* const* ptr;
    const uint64_t* ends;
    size_t n;
    // fn(src, dst_offset, len) for every stretch of [a, a + len) that lies in one piece, in order
    template <class Fn>
    void segments(uint64_t a, uint64_t len, Fn&& fn) const {
        if (!ends) { fn(ptr[0] + a, (uint64_t)0, len); return; }
        size_t i = (size_t)(std::upper_bound(ends, ends + n, a) - ends);
        uint64_t at = 0;
        while (at < len && i < n) {
            const uint64_t p0 = i ? ends[i - 1] : 0, take = std::min(len - at, ends[i] - (a + at));
            if (take) fn(ptr[i] + (a + at - p0), at, take);
            at += take;
            i++;
        }
    }
};

static int classify_records_parts(umgap_index* ix, umgap_taxonomy* tx, int method, int strategy, float factor, float lower_bound,
                                  int ranked_only, const ByteParts bytes, const uint64_t* frag_offsets, size_t n_fragments,
                                  const uint64_t* read_frags, size_t n_reads, int seedextend, uint32_t min_seed_size,
                                  uint32_t max_gap_size, uint32_t* out_taxon, uint32_t* out_status) {
    if (!ix || !tx || (n_reads && (!read_frags || !out_taxon || !out_status)) || (n_fragments && !frag_offsets))
        return fail(UMGAP_ERR_INVALID, "umgap_classify_records: null argument");
    // a table bound to THIS taxonomy (umgap_index_bind_taxonomy: values = preorder interval codes) is what the fused route wants -- no
    // by-id gathers in the aggregate; the general route turns the codes back into taxon ids behind its probe (k_codes_to_ids)
    if (ix->bound_to && ix->bound_to != tx) return fail(UMGAP_ERR_INVALID, "umgap_classify_records: the index is bound to another taxonomy");
    const bool bound = ix->bound_to != nullptr;
    if (ix->kind != 9 && ix->kind != 10) return fail(UMGAP_ERR_INVALID, "umgap_classify_records: not a 9-mer index");
    const bool walk = ix->kind == 10;   // window numbers -> taxa through the walk array (index_walk.hip) before the hits are used
    if (ix->device != tx->device) return fail(UMGAP_ERR_INVALID, "umgap_classify_records: index and taxonomy live on different devices");
    if (method != UMGAP_METHOD_TREE && method != UMGAP_METHOD_RMQ) return fail(UMGAP_ERR_INVALID, "unknown method");
    if (strategy < 0 || strategy > 2) return fail(UMGAP_ERR_INVALID, "unknown strategy");
    if (method == UMGAP_METHOD_TREE && strategy == UMGAP_STRATEGY_MRTL)
        return fail(UMGAP_ERR_INVALID, "Invalid invocation: Tree and MaximumRootToLeafPath cannot be combined");  // src/commands/taxa2agg.rs:134-138
    if (!n_reads) return UMGAP_OK;
    if (read_frags[0] != 0 || read_frags[n_reads] != n_fragments) return fail(UMGAP_ERR_INVALID, "umgap_classify_records: read_frags must cover [0, n_fragments]");
    int rc = use_device(tx->device);
    if (rc) return rc;
    const uint64_t base = n_fragments ? frag_offsets[0] : 0;
    if (n_fragments && frag_offsets[n_fragments] < base) return fail(UMGAP_ERR_INVALID, "umgap_classify_records: frag_offsets must not decrease");
    const uint64_t n_bytes = n_fragments ? frag_offsets[n_fragments] - base : 0;
    if (n_bytes) {
        if (!bytes.ptr || !bytes.n) return fail(UMGAP_ERR_INVALID, "umgap_classify_records: null bytes");
        for (size_t i = 0; i < bytes.n; i++) {
            const uint64_t p0 = i && bytes.ends ? bytes.ends[i - 1] : 0;
            if (bytes.ends && bytes.ends[i] < p0) return fail(UMGAP_ERR_INVALID, "umgap_classify_records_sg: piece_ends must not decrease");
            if (!bytes.ptr[i] && (!bytes.ends || bytes.ends[i] > p0)) return fail(UMGAP_ERR_INVALID, "umgap_classify_records: null bytes");
        }
        if (bytes.ends && bytes.ends[bytes.n - 1] < base + n_bytes) return fail(UMGAP_ERR_INVALID, "umgap_classify_records_sg: the pieces end before the last fragment does");
    }

    // the fused route (k_records_fit): every read of F fragments, F from the first read and the totals; UMGAP_RECORDS_FUSED=0 keeps
    // the general kernels
    uint32_t F_all = 0;
    if (n_reads && read_frags[1] >= 1 && read_frags[1] <= PROT_LONG_FRAGS && n_fragments == (uint64_t)read_frags[1] * n_reads) F_all = (uint32_t)read_frags[1];
    if (const char* env = getenv("UMGAP_RECORDS_FUSED")) if (env[0] == '0') F_all = 0;
    bool unfit = false;
    auto run = [&](const bool fused) -> int {
    // ---- chunk plan: whole reads, at most CH residues / FH fragments / RH reads each (a single read may exceed them) ----
    uint64_t CH = 32ull << 20;
    if (const char* env = getenv("UMGAP_RECORDS_CHUNK_MB")) { const long v = atol(env); if (v >= 1 && v <= 1024) CH = (uint64_t)v << 20; }
    const size_t FH = 4u << 20, RH = 2u << 20;
    auto frag_of = [&](size_t r) -> uint64_t { uint64_t f = read_frags[r]; return f > n_fragments ? n_fragments : f; };
    auto byte_of = [&](size_t r) -> uint64_t { uint64_t o = frag_offsets[frag_of(r)]; return o < base ? 0 : std::min(o - base, n_bytes); };
    std::vector<size_t> cuts{0};
    while (cuts.back() < n_reads) {
        const size_t r0 = cuts.back();
        size_t lo = r0 + 1, hi = std::min(n_reads, r0 + RH);   // largest r1 in [lo, hi] within the budgets (monotone if the input is sane)
        const uint64_t b0 = byte_of(r0), f0 = frag_of(r0);
        auto fits = [&](size_t r1) { uint64_t b = byte_of(r1), f = frag_of(r1); return b >= b0 && b - b0 <= CH && f >= f0 && f - f0 <= FH; };
        while (lo < hi) {
            size_t mid = lo + (hi - lo + 1) / 2;
            if (fits(mid)) lo = mid; else hi = mid - 1;
        }
        cuts.push_back(lo);
    }
    uint64_t max_b = 0;
    size_t max_f = 0, max_r = 0;
    for (size_t c = 0; c + 1 < cuts.size(); c++) {
        const uint64_t b0 = byte_of(cuts[c]), b1 = byte_of(cuts[c + 1]), f0 = frag_of(cuts[c]), f1 = frag_of(cuts[c + 1]);
        if (b1 < b0 || f1 < f0) return fail(UMGAP_ERR_INVALID, "umgap_classify_records: offsets must not decrease");
        if (b1 - b0 >= 0xFFFFFF00ull) return fail(UMGAP_ERR_INVALID, "umgap_classify_records: a single read holds 4 GiB of residues or more");
        max_b = std::max(max_b, b1 - b0);
        max_f = std::max<size_t>(max_f, f1 - f0);
        max_r = std::max(max_r, cuts[c + 1] - cuts[c]);
    }
    const bool want_kept = seedextend && min_seed_size < 2;   // the only settings under which the reference's seedextend can panic

    // ---- device staging: two slots inside the taxonomy handle's staging block ----
    const size_t b_bytes = pad256(max_b + 160), b_off = pad256((max_f + 1) * 8), b_rf = pad256((max_r + 1) * 8), b_f32 = pad256((max_f + 1) * 4),
                 b_tails = pad256((max_b / 32 + 8) * 4), b_hits = pad256((max_b + 16) * 4), b_out = pad256(max_r * 4 + 4);
    const size_t per_slot = b_bytes + 2 * b_off + 3 * b_rf + 2 * b_f32 + b_tails + b_hits + (want_kept ? b_f32 : 0) + 2 * b_out + 256;
    constexpr int NS = 3;   // staging slots: the uploads run up to two chunks ahead of the kernels
    const size_t need = NS * per_slot;
    if (need > tx->d_stage_bytes) {
        if (tx->d_stage) (void)hipFree(tx->d_stage);
        tx->d_stage = nullptr;
        tx->d_stage_bytes = 0;
        hipError_t e = hipMalloc(&tx->d_stage, need + need / 8);
        if (e != hipSuccess) return fail(UMGAP_ERR_NOMEM, std::string("hipMalloc(record staging): ") + hipGetErrorString(e));
        tx->d_stage_bytes = need + need / 8;
    }
    // callers with pageable arrays: the chunks go through pinned staging with a threaded memcpy (host_copy.hpp), the results come
    // back through it too -- an asynchronous copy from / to pageable memory blocks the host until it is done, which would keep the
    // upload lane from running ahead
    auto is_pinned = [](const void* p0, const void* p1) {
        hipPointerAttribute_t a{};
        for (const void* p : {p0, p1}) {
            if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
            if (a.type != hipMemoryTypeHost) return false;
        }
        return true;
    };
    bool in_pinned = (!n_fragments || is_pinned(frag_offsets, frag_offsets + n_fragments)) && is_pinned(read_frags, read_frags + n_reads);
    if (in_pinned && n_bytes) bytes.segments(base, n_bytes, [&](const uint8_t* src, uint64_t, uint64_t len) { in_pinned = in_pinned && is_pinned(src, src + len - 1); });
    const bool out_pinned = is_pinned(out_taxon, out_taxon + n_reads - 1) && is_pinned(out_status, out_status + n_reads - 1);
    const size_t hb_bytes = pad256(max_b + 16), hb_off = pad256((max_f + 1) * 8), hb_rf = pad256((max_r + 1) * 8), hb_out = pad256(max_r * 4 + 4);
    const size_t h_per_slot = (in_pinned ? 0 : hb_bytes + hb_off + hb_rf) + (out_pinned ? 0 : 2 * hb_out);
    if (NS * h_per_slot > tx->h_stage_bytes) {
        if (tx->h_stage) (void)hipHostFree(tx->h_stage);
        tx->h_stage = nullptr;
        tx->h_stage_bytes = 0;
        if (hipHostMalloc(&tx->h_stage, NS * h_per_slot, hipHostMallocDefault) != hipSuccess) return fail(UMGAP_ERR_NOMEM, "umgap_classify_records: hipHostMalloc of the staging failed");
        tx->h_stage_bytes = NS * h_per_slot;
    }
    struct HostSlot { uint8_t* bytes = nullptr; uint64_t *off = nullptr, *rf = nullptr; uint32_t *ot = nullptr, *os = nullptr; } H[NS];
    for (int k = 0; k < NS; k++) {
        char* p = (char*)tx->h_stage + (size_t)k * h_per_slot;
        if (!in_pinned) { H[k].bytes = (uint8_t*)p; p += hb_bytes; H[k].off = (uint64_t*)p; p += hb_off; H[k].rf = (uint64_t*)p; p += hb_rf; }
        if (!out_pinned) { H[k].ot = (uint32_t*)p; p += hb_out; H[k].os = (uint32_t*)p; }
    }
    size_t out_at[NS] = {}, out_n[NS] = {};       // results of the slot's chunk still in the staging (pageable outputs)
    auto drain_out = [&](int k) {
        if (!out_n[k]) return;
        memcpy(out_taxon + out_at[k], H[k].ot, out_n[k] * 4);
        memcpy(out_status + out_at[k], H[k].os, out_n[k] * 4);
        out_n[k] = 0;
    };
    Slot S[NS];
    hipError_t e = hipSuccess;
    if (int lanes = ensure_lanes(tx)) return lanes;
    // an UPLOAD lane and a COMPUTE lane, tied by events per slot: with one lane per chunk (rounds 2-3) the two lanes fell into step --
    // both uploading, then both computing -- and the link idled while the kernels ran (rocprofv3 --memory-copy-trace: 2.2 ms per
    // two chunks instead of 1.4)
    const hipStream_t up = tx->stream2, run_on = tx->stream3;
    if (!tx->h_flags) e = hipHostMalloc((void**)&tx->h_flags, 4 * NS * sizeof(uint32_t), hipHostMallocDefault);
    if (e != hipSuccess) { tx->h_flags = nullptr; return fail(UMGAP_ERR_NOMEM, std::string("hipHostMalloc: ") + hipGetErrorString(e)); }
    volatile uint32_t* h_bad = tx->h_flags;    // per slot: {offsets bad, a read unfit for the short fused kernel, unfit for the LONG one, a read left UNSUPPORTED}
    for (int k = 0; k < 4 * NS; k++) h_bad[k] = 0;
    struct Events {
        hipEvent_t uploaded[NS] = {}, done[NS] = {};     // per slot: the chunk's arrays are on the device / its results have left
        ~Events() { for (hipEvent_t x : uploaded) if (x) (void)hipEventDestroy(x); for (hipEvent_t x : done) if (x) (void)hipEventDestroy(x); }
    } ev;
    for (int k = 0; k < NS; k++) {
        UMGAP_HIP(hipEventCreateWithFlags(&ev.uploaded[k], hipEventDisableTiming));
        UMGAP_HIP(hipEventCreateWithFlags(&ev.done[k], hipEventDisableTiming));
    }
    for (int k = 0; k < NS; k++) {
        char* p = (char*)tx->d_stage + (size_t)k * per_slot;
        S[k].bytes = (uint8_t*)p; p += b_bytes;
        S[k].off = (uint64_t*)p; p += b_off;
        S[k].off_rel = (uint64_t*)p; p += b_off;
        S[k].rf = (uint64_t*)p; p += b_rf;
        S[k].roff = (uint64_t*)p; p += b_rf;
        S[k].rfl = (uint64_t*)p; p += b_rf;
        S[k].fstart = (uint32_t*)p; p += b_f32;
        S[k].flen = (uint32_t*)p; p += b_f32;
        S[k].tails = (uint32_t*)p; p += b_tails;
        S[k].hits = (uint32_t*)p; p += b_hits;
        if (want_kept) { S[k].kept = (uint32_t*)p; p += b_f32; }
        S[k].ot = (uint32_t*)p; p += b_out;
        S[k].os = (uint32_t*)p; p += b_out;
        S[k].bad = (uint32_t*)p;
    }
    TableView T = ix->view;
    rc = UMGAP_OK;
    for (size_t c = 0; c + 1 < cuts.size() && rc == UMGAP_OK; c++) {
        const int slot = (int)(c % NS);
        Slot& L = S[slot];
        const size_t r0 = cuts[c], nr = cuts[c + 1] - r0;
        const uint64_t f0 = frag_of(r0), nf = frag_of(cuts[c + 1]) - f0;
        const uint64_t b0 = byte_of(r0), nb = byte_of(cuts[c + 1]) - b0;
        // ---- upload lane: COPIES ONLY.  The slot's previous tenant (chunk c - NS) is through: the host waited for it below.  No fill,
        // no wait on an event here: those are packets of a compute queue, the runtime folds a process's streams onto four of them
        // (tools/native/queue_probe.hip: streams 0..9 -> queues 0 1 2 3 3 2 1 0 3 2), and on a queue shared with the compute lane such
        // a packet -- and every upload behind it -- waits for the running kernel ----
        const uint64_t *src_o = frag_offsets + f0, *src_r = read_frags + r0;
        hipError_t e_up = hipSuccess;
        if (!in_pinned) {          // (the slot's previous upload is long through: the host waited for that chunk's results below)
            if (nb) bytes.segments(base + b0, nb, [&](const uint8_t* src, uint64_t at, uint64_t len) { par_memcpy(H[slot].bytes + at, src, len); });
            if (nf) par_memcpy(H[slot].off, src_o, (nf + 1) * 8);
            par_memcpy(H[slot].rf, src_r, (nr + 1) * 8);
            src_o = H[slot].off; src_r = H[slot].rf;
            if (nb) e_up = hipMemcpyAsync(L.bytes, H[slot].bytes, nb, hipMemcpyHostToDevice, up);
        } else if (nb)             // pinned pieces are copied from where they lie, one copy per piece the chunk touches
            bytes.segments(base + b0, nb, [&](const uint8_t* src, uint64_t at, uint64_t len) {
                if (e_up == hipSuccess) e_up = hipMemcpyAsync(L.bytes + at, src, len, hipMemcpyHostToDevice, up);
            });
        UMGAP_HIP_BREAK(e_up);
        if (nf) UMGAP_HIP_BREAK(hipMemcpyAsync(L.off, src_o, (nf + 1) * 8, hipMemcpyHostToDevice, up));
        UMGAP_HIP_BREAK(hipMemcpyAsync(L.rf, src_r, (nr + 1) * 8, hipMemcpyHostToDevice, up));
        UMGAP_HIP_BREAK(hipEventRecord(ev.uploaded[slot], up));
        // ---- compute lane ----
        const hipStream_t s = run_on;
        UMGAP_HIP_BREAK(hipStreamWaitEvent(s, ev.uploaded[slot], 0));
        UMGAP_HIP_BREAK(hipMemsetAsync(L.bad, 0, 16, s));
        if (!fused) UMGAP_HIP_BREAK(hipMemsetAsync(L.tails, 0, (nb / 32 + 8) * 4, s));
        UMGAP_HIP_BREAK(hipMemsetAsync(L.bytes + nb, 0, 144, s));   // the packer may look 8 bytes past the chunk
        const unsigned gf = (unsigned)std::min<size_t>((nf + 255) / 256 + 1, 2048), gr = (unsigned)std::min<size_t>((nr + 256) / 256 + 1, 2048);
        if (fused) {
            if (!nf) UMGAP_HIP_BREAK(hipMemsetAsync(L.off, 0, 8, s));
            hipLaunchKernelGGL(k_records_fit, dim3(std::max(gf, gr)), dim3(256), 0, s, (const uint64_t*)L.off, (size_t)nf, (const uint64_t*)L.rf, nr, f0, F_all, base + b0,
                               (uint32_t)nb, L.off_rel, L.bad);
            UMGAP_HIP_BREAK(hipGetLastError());
            if (F_all <= PROT_MAX_FRAGS) {
                rc = classify_prot9_gated(ix, tx, method, strategy, factor, lower_bound, ranked_only, L.bytes, L.off_rel, nr, F_all, seedextend, min_seed_size, max_gap_size, L.ot, L.os,
                                          L.bad, 0, s);
                if (rc) break;
            }
            rc = classify_prot9_gated(ix, tx, method, strategy, factor, lower_bound, ranked_only, L.bytes, L.off_rel, nr, F_all, seedextend, min_seed_size, max_gap_size, L.ot, L.os,
                                      L.bad, 1, s);
            if (rc) break;
            hipLaunchKernelGGL(k_records_any_unsupported, dim3(gr), dim3(256), 0, s, (const uint32_t*)L.os, nr, L.bad);
            UMGAP_HIP_BREAK(hipGetLastError());
        } else {
        if (nf) hipLaunchKernelGGL(k_frag_prepare, dim3(gf), dim3(256), 0, s, L.off, (size_t)nf, base + b0, (uint32_t)nb, L.fstart, L.flen, L.tails, L.bad);
        if (nb) {
            const unsigned gb = (unsigned)std::min<uint64_t>((nb + 511) / 512, 2048);
            hipLaunchKernelGGL(k_lookup_bytes9, dim3(gb), dim3(256), 0, s, T, L.bytes, L.tails, (uint32_t)nb, (uint32_t)nb, L.hits, 0u, walk ? walk_ref_of(ix) : WalkRef{});
            if (bound) hipLaunchKernelGGL(k_codes_to_ids, dim3(gb), dim3(256), 0, s, L.hits, (size_t)nb, (const uint32_t*)tx->d_id_of_pre);
        }
        if (seedextend && nf)
            hipLaunchKernelGGL(k_seedextend_frags, dim3(gf), dim3(256), 0, s, L.fstart, L.flen, (size_t)nf, min_seed_size, max_gap_size, L.hits, L.kept);
        hipLaunchKernelGGL(k_read_offsets, dim3(gr), dim3(256), 0, s, L.rf, nr, f0, (size_t)nf, L.fstart, (uint32_t)nb, L.roff, L.rfl, L.bad);
        UMGAP_HIP_BREAK(hipGetLastError());
        rc = umgap_aggregate_dev(tx, method, strategy, factor, lower_bound, ranked_only, L.hits, nullptr, L.roff, nr, L.ot, L.os, s);
        if (!rc && want_kept && nf) rc = mark_seedextend_panics(L.kept, L.rfl, nr, L.ot, L.os, s);
        if (rc) break;
        }
        UMGAP_HIP_BREAK(hipMemcpyAsync(out_pinned ? out_taxon + r0 : H[slot].ot, L.ot, nr * 4, hipMemcpyDeviceToHost, s));
        UMGAP_HIP_BREAK(hipMemcpyAsync(out_pinned ? out_status + r0 : H[slot].os, L.os, nr * 4, hipMemcpyDeviceToHost, s));
        if (!out_pinned) { out_at[slot] = r0; out_n[slot] = nr; }
        UMGAP_HIP_BREAK(hipMemcpyAsync(tx->h_flags + 4 * slot, L.bad, 16, hipMemcpyDeviceToHost, s));
        UMGAP_HIP_BREAK(hipEventRecord(ev.done[slot], s));
        // the host stays at most NS chunks ahead, and looks at the flags of the chunk whose slot the next one takes
        if (c + 1 >= (size_t)NS) {
            const int w = (int)((c + 1 - NS) % NS);
            UMGAP_HIP_BREAK(hipEventSynchronize(ev.done[w]));
            drain_out(w);
            if (h_bad[4 * w]) { rc = fail(UMGAP_ERR_INVALID, "umgap_classify_records: frag_offsets / read_frags are not non-decreasing offsets into the block"); break; }
            if (fused && (h_bad[4 * w + 2] || h_bad[4 * w + 3])) { unfit = true; break; }    // a read neither variant of the fused kernel takes: the caller runs the general kernels
        }
    }
    hipError_t e0 = hipStreamSynchronize(up), e1 = hipStreamSynchronize(run_on);
    if (!rc && !unfit && e0 == hipSuccess && e1 == hipSuccess) for (int k = 0; k < NS; k++) drain_out(k);
    if (rc) return rc;
    if (e0 != hipSuccess || e1 != hipSuccess) return fail(UMGAP_ERR_DEVICE, std::string("umgap_classify_records: ") + hipGetErrorString(e0 != hipSuccess ? e0 : e1));
    for (int k = 0; k < NS; k++) {
        if (h_bad[4 * k]) return fail(UMGAP_ERR_INVALID, "umgap_classify_records: frag_offsets / read_frags are not non-decreasing offsets into the block");
        if (fused && (h_bad[4 * k + 2] || h_bad[4 * k + 3])) unfit = true;
    }
    return UMGAP_OK;
    };   // run
    if (F_all) {
        g_last_route = 1;
        rc = run(true);
        if (rc || !unfit) return rc;
        g_last_route = 2;
        unfit = false;
    } else g_last_route = 0;
    return run(false);
}

extern "C" int umgap_classify_records(umgap_index* ix, umgap_taxonomy* tx, int method, int strategy, float factor, float lower_bound,
                                      int ranked_only, const uint8_t* bytes, const uint64_t* frag_offsets, size_t n_fragments,
                                      const uint64_t* read_frags, size_t n_reads, int seedextend, uint32_t min_seed_size,
                                      uint32_t max_gap_size, uint32_t* out_taxon, uint32_t* out_status) {
    return classify_records_parts(ix, tx, method, strategy, factor, lower_bound, ranked_only, ByteParts{&bytes, nullptr, bytes ? (size_t)1 : (size_t)0}, frag_offsets, n_fragments,
                                  read_frags, n_reads, seedextend, min_seed_size, max_gap_size, out_taxon, out_status);
}

extern "C" int umgap_classify_records_sg(umgap_index* ix, umgap_taxonomy* tx, int method, int strategy, float factor, float lower_bound,
                                         int ranked_only, const uint8_t* const* pieces, const uint64_t* piece_ends, size_t n_pieces,
                                         const uint64_t* frag_offsets, size_t n_fragments, const uint64_t* read_frags, size_t n_reads,
                                         int seedextend, uint32_t min_seed_size, uint32_t max_gap_size, uint32_t* out_taxon,
                                         uint32_t* out_status) {
    if (n_pieces && (!pieces || !piece_ends)) return fail(UMGAP_ERR_INVALID, "umgap_classify_records_sg: null argument");
    return classify_records_parts(ix, tx, method, strategy, factor, lower_bound, ranked_only, ByteParts{pieces, piece_ends, n_pieces}, frag_offsets, n_fragments,
                                  read_frags, n_reads, seedextend, min_seed_size, max_gap_size, out_taxon, out_status);
}

// which kernels the last umgap_classify_records of this thread ran: 0 the general ones, 1 the fused one, 2 the fused one and then,
// a read not fitting it, the general ones (tests and benchmarks)
extern "C" int umgap_records_last_route(void) { return g_last_route; }
