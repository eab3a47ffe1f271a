// pipeline.hip — library plumbing (errors, device selection), the seedextend stage on device and
// the synthetic read generator used by bench.py / the parity tests.  gfx950 only.
#include "taxonomy.hpp"

namespace umgap {

static thread_local std::string g_last_error;

void set_error(const std::string& msg) { g_last_error = msg; }
int fail(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}

int use_device(int device) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(UMGAP_ERR_DEVICE, "no usable GPU (hipGetDeviceCount: " + std::string(e == hipSuccess ? "0 devices" : hipGetErrorString(e)) +
                                          "); libumgap_hip has no CPU fallback");
    if (device < 0 || device >= count) return fail(UMGAP_ERR_INVALID, "device " + std::to_string(device) + " out of range (" + std::to_string(count) + " visible)");
    e = hipSetDevice(device);
    if (e != hipSuccess) return fail(UMGAP_ERR_DEVICE, std::string("hipSetDevice: ") + hipGetErrorString(e));
    return UMGAP_OK;
}

}  // namespace umgap

#include "seedextend.hpp"

using namespace umgap;

namespace {

// ---------------------------------------------------------------------------------------------
// seedextend, unranked mode (seedextend.hpp has the per-record state machine), in place, one lane per record.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_seedextend(const uint64_t* __restrict__ offsets, uint32_t stride, size_t n_records,
                                                    uint32_t min_seed, uint32_t max_gap, uint32_t* __restrict__ taxa,
                                                    uint32_t* __restrict__ kept) {
    size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (; r < n_records; r += step) {
        uint64_t off = offsets ? offsets[r] : (uint64_t)r * stride;
        uint32_t n = offsets ? (uint32_t)(offsets[r + 1] - off) : stride;
        const uint32_t k = seedextend_record(taxa + off, n, min_seed, max_gap);
        if (kept) kept[r] = k;
    }
}

// ranked mode (`seedextend -r`): only the best-scoring extended seed of a record stays
__global__ __launch_bounds__(256) void k_seedextend_ranked(const uint64_t* __restrict__ offsets, size_t n_records, uint32_t min_seed, uint32_t max_gap,
                                                           const int8_t* __restrict__ scores, uint32_t n_scores, uint32_t penalty, uint32_t* __restrict__ taxa,
                                                           uint32_t* __restrict__ kept) {
    size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (; r < n_records; r += step) {
        const uint64_t off = offsets[r];
        const uint32_t k = seedextend_record_ranked(taxa + off, (uint32_t)(offsets[r + 1] - off), min_seed, max_gap, scores, n_scores, penalty);
        if (kept) kept[r] = k;
    }
}

// Reads that hold a fragment on which seedextend panicked (kept == UMGAP_NONE) get status UMGAP_READ_PANIC: in the
// reference's pipe the seedextend process dies at that record (exit 101) and nothing after it is printed.
__global__ __launch_bounds__(256) void k_mark_seed_panics(const uint32_t* __restrict__ kept, const uint64_t* __restrict__ read_frags, size_t n_reads,
                                                          uint32_t* __restrict__ out_taxon, uint32_t* __restrict__ out_status) {
    size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (; r < n_reads; r += step) {
        bool pk = false;
        for (uint64_t f = read_frags[r]; f < read_frags[r + 1]; f++) pk = pk || kept[f] == UMGAP_NONE;
        if (pk) { out_taxon[r] = 0; out_status[r] = UMGAP_READ_PANIC; }
    }
}

// ---------------------------------------------------------------------------------------------
// synthetic reads (DESIGN.md "Synthetic workload", seed 44)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__device__ __forceinline__ uint64_t synth_key(uint64_t i, uint64_t a, uint64_t b) {
    const uint64_t M = 512000000000ull;
    uint64_t x = (a * i + b) % M;
    uint64_t key = 0;
#pragma unroll
    for (int j = 0; j < 9; j++) {
        uint32_t d = (uint32_t)(x % 20);
        x /= 20;
        uint32_t code = d + (d >= 1) + (d >= 8) + (d >= 12) + (d >= 17) + (d >= 19);
        key |= (uint64_t)code << (5 * j);
    }
    return key;
}

__global__ __launch_bounds__(256) void k_synth_reads(TaxView V, const uint32_t* __restrict__ taxon_of_row,
                                                     const uint32_t* __restrict__ row_of_id, uint64_t n_tax, uint64_t n_keys,
                                                     uint64_t a, uint64_t b, uint64_t seed, uint64_t first_pair,
                                                     uint64_t n_pairs, uint32_t kpp, uint32_t path_pct, uint32_t unrel_pct,
                                                     uint64_t* __restrict__ keys) {
    const uint64_t M = 512000000000ull;
    uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t step = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t total = n_pairs * kpp;
    for (; g < total; g += step) {
        uint64_t p = first_pair + g / kpp;
        uint32_t j = (uint32_t)(g % kpp);
        uint64_t r0 = splitmix64(seed ^ (p * 0x9E3779B97F4A7C15ull));
        uint64_t s = r0 % n_tax;  // source taxon (row)
        uint64_t rj = splitmix64(r0 + (uint64_t)(j + 1) * 0xD1B54A32D192ED03ull);
        uint32_t cls = (uint32_t)(rj % 100);
        uint64_t member = ~0ull;
        if (cls < path_pct) {
            // member whose taxon lies on the source's root path: half the time the source itself
            uint32_t pre = V.by_id[taxon_of_row[s]].x;
            uint32_t depth = V.depth[pre];
            uint32_t up = ((rj >> 8) & 1) ? (uint32_t)((rj >> 9) % (depth + 1)) : 0u;
            for (uint32_t k = 0; k < up; k++) pre = V.parent_pre[pre];
            uint64_t ra = row_of_id[V.id_of_pre[pre]];
            if (ra < n_keys) {
                uint64_t cnt = (n_keys - ra + n_tax - 1) / n_tax;
                member = ra + n_tax * ((rj >> 20) % cnt);
            }
        } else if (cls < path_pct + unrel_pct) {
            // unrelated member; slots j and j^1 share it, so some unrelated taxa occur twice and
            // survive `-l 2`
            uint64_t rp = splitmix64(r0 + (uint64_t)((j >> 1) + 1) * 0xA24BAED4963EE407ull);
            member = rp % n_keys;
        }
        uint64_t key;
        if (member != ~0ull) {
            key = synth_key(member, a, b);
        } else {
            key = synth_key(n_keys + (rj >> 8) % (M - n_keys), a, b);  // pi(i), i >= n_keys: never a member
            if (cls == 99) key = (key & ~(31ull << 10)) | (23ull << 10);  // an 'X': outside the 20-letter alphabet
        }
        keys[g] = key;
    }
}

// synthetic tryptic-like peptides: peptide i has length offsets[i+1]-offsets[i] and letters drawn
// from the 20 amino acids by a hash of (seed, i, position); the last letter is K or R
__global__ __launch_bounds__(256) void k_synth_peptides(uint64_t first, size_t n, uint64_t seed, const uint64_t* __restrict__ offsets,
                                                        uint8_t* __restrict__ bytes) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += step) {
        uint64_t o = offsets[i], len = offsets[i + 1] - o;
        uint64_t h = splitmix64(seed ^ ((first + i) * 0x9E3779B97F4A7C15ull));
        for (uint64_t j = 0; j < len; j++) {
            if ((j & 7) == 0 && j) h = splitmix64(h + j);
            uint32_t d = (uint32_t)((h >> (8 * (j & 7))) & 0xFF) % 20;
            uint32_t code = d + (d >= 1) + (d >= 8) + (d >= 12) + (d >= 17) + (d >= 19);
            bytes[o + j] = (uint8_t)('A' + code);
        }
        if (len) bytes[o + len - 1] = (h >> 60) & 1 ? 'K' : 'R';
    }
}

// synthetic peptides that a tryptic digest gives back whole: no K or R inside, no leading P, the last letter K or R.  Peptide i
// has global number ids[i] (or first + i) -- a read for the fused tryptic pipeline is a list of such numbers, written back to back.
__global__ __launch_bounds__(256) void k_synth_tryptic(const uint64_t* __restrict__ ids, uint64_t first, size_t n, uint64_t seed,
                                                       const uint64_t* __restrict__ offsets, uint8_t* __restrict__ bytes) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += step) {
        const uint64_t o = offsets[i], len = offsets[i + 1] - o, number = ids ? ids[i] : first + i;
        uint64_t h = splitmix64(seed ^ (number * 0x9E3779B97F4A7C15ull));
        for (uint64_t j = 0; j < len; j++) {
            if ((j & 7) == 0 && j) h = splitmix64(h + j);
            uint32_t d = (uint32_t)((h >> (8 * (j & 7))) & 0xFF) % 18;
            if (j == 0 && d == 11) d = 12;                                                     // no leading P
            bytes[o + j] = (uint8_t)"ACDEFGHILMNPQSTVWY"[d];
        }
        if (len) bytes[o + len - 1] = (h >> 60) & 1 ? 'K' : 'R';
    }
}

// synthetic surface-B reads: read r = F fragments of `flen` residues, each a copy of a proteome
// fragment of the read's source taxon family with point mutations (a mutated residue makes the up to
// nine windows that cover it miss); 1 % of the residues become '*' (stop codon, not [A-Z]).
__global__ __launch_bounds__(256) void k_synth_prot_reads(const uint8_t* __restrict__ proteome, uint64_t n_prot_frag, uint32_t flen, uint64_t seed,
                                                          uint64_t first_read, size_t n_reads, uint32_t F, uint32_t mut_per_mille,
                                                          uint8_t* __restrict__ out) {
    size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x, total = n_reads * F * flen;
    for (; g < total; g += step) {
        size_t r = g / ((size_t)F * flen);
        uint32_t f = (uint32_t)((g / flen) % F), c = (uint32_t)(g % flen);
        uint64_t r0 = splitmix64(seed ^ ((first_read + r) * 0x9E3779B97F4A7C15ull));
        // the read's fragments come from neighbouring proteome fragments (same or nearby taxa)
        uint64_t src = (r0 + (uint64_t)f * 0x632BE59BD9B4E019ull % 5) % n_prot_frag;
        uint8_t ch = proteome[src * flen + c];
        uint64_t hm = splitmix64(r0 ^ ((uint64_t)(f * flen + c + 1) * 0xD1B54A32D192ED03ull));
        uint32_t roll = (uint32_t)(hm % 1000);
        if (roll < mut_per_mille) {
            uint32_t d = (uint32_t)((hm >> 20) % 20);
            ch = roll < 10 ? (uint8_t)'*' : (uint8_t)('A' + d + (d >= 1) + (d >= 8) + (d >= 12) + (d >= 17) + (d >= 19));
        }
        out[g] = ch;
    }
}

inline int grid_for(size_t n) {
    size_t g = (n + 255) / 256;
    if (g < 1) g = 1;
    if (g > 256 * 16) g = 256 * 16;
    return (int)g;
}

}  // namespace

namespace umgap {
int mark_seedextend_panics(const uint32_t* d_kept, const uint64_t* d_read_frags, size_t n_reads, uint32_t* d_out_taxon, uint32_t* d_out_status,
                           hipStream_t stream) {
    if (!n_reads) return UMGAP_OK;
    size_t g = (n_reads + 255) / 256;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(k_mark_seed_panics, dim3((unsigned)g), dim3(256), 0, stream, d_kept, d_read_frags, n_reads, d_out_taxon, d_out_status);
    UMGAP_HIP(hipGetLastError());
    return UMGAP_OK;
}
}  // namespace umgap

extern "C" {

const char* umgap_last_error(void) { return g_last_error.c_str(); }
const char* umgap_version(void) { return "1.1.1"; }
int umgap_device_count(void) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) return 0;
    return count;
}

int umgap_seedextend_dev(const uint64_t* d_offsets, size_t n_records, uint32_t min_seed_size, uint32_t max_gap_size,
                         uint32_t* d_taxa, uint32_t* d_kept, int device, void* stream) {
    if (n_records && (!d_offsets || !d_taxa)) return fail(UMGAP_ERR_INVALID, "umgap_seedextend_dev: null argument");
    int rc = use_device(device);
    if (rc) return rc;
    if (n_records) hipLaunchKernelGGL(k_seedextend, dim3(grid_for(n_records)), dim3(256), 0, (hipStream_t)stream, d_offsets, 0u, n_records, min_seed_size, max_gap_size, d_taxa, d_kept);
    UMGAP_HIP(hipGetLastError());
    return UMGAP_OK;
}

int umgap_seedextend_ranked_dev(umgap_taxonomy* tx, const uint64_t* d_offsets, size_t n_records, uint32_t min_seed_size, uint32_t max_gap_size,
                                uint32_t penalty, uint32_t* d_taxa, uint32_t* d_kept, void* stream) {
    if (!tx || (n_records && (!d_offsets || !d_taxa))) return fail(UMGAP_ERR_INVALID, "umgap_seedextend_ranked_dev: null argument");
    int rc = use_device(tx->device);
    if (rc) return rc;
    {                                   // the per-id scores go up once (callers on several threads meet here)
        std::lock_guard<std::mutex> g(tx->defer_mutex);
        if (!tx->d_seed_score) {
            int8_t* d = nullptr;
            UMGAP_HIP(hipMalloc((void**)&d, tx->h_seed_score.size()));
            if (hipMemcpy(d, tx->h_seed_score.data(), tx->h_seed_score.size(), hipMemcpyHostToDevice) != hipSuccess) {
                (void)hipFree(d);
                return fail(UMGAP_ERR_DEVICE, "umgap_seedextend_ranked_dev: copy of the scores failed");
            }
            tx->d_seed_score = d;
        }
    }
    if (n_records) hipLaunchKernelGGL(k_seedextend_ranked, dim3(grid_for(n_records)), dim3(256), 0, (hipStream_t)stream, d_offsets, n_records, min_seed_size, max_gap_size,
                                      (const int8_t*)tx->d_seed_score, tx->len, penalty, d_taxa, d_kept);
    UMGAP_HIP(hipGetLastError());
    return UMGAP_OK;
}

int umgap_seedextend_strided_dev(uint32_t stride, size_t n_records, uint32_t min_seed_size, uint32_t max_gap_size,
                                 uint32_t* d_taxa, uint32_t* d_kept, int device, void* stream) {
    if (n_records && !d_taxa) return fail(UMGAP_ERR_INVALID, "umgap_seedextend_strided_dev: null argument");
    int rc = use_device(device);
    if (rc) return rc;
    if (n_records) hipLaunchKernelGGL(k_seedextend, dim3(grid_for(n_records)), dim3(256), 0, (hipStream_t)stream, (const uint64_t*)nullptr, stride, n_records, min_seed_size, max_gap_size, d_taxa, d_kept);
    UMGAP_HIP(hipGetLastError());
    return UMGAP_OK;
}

int umgap_synth_peptides_dev(uint64_t first, size_t n, uint64_t seed, const uint64_t* d_offsets, uint8_t* d_bytes, int device, void* stream) {
    if (n && (!d_offsets || !d_bytes)) return fail(UMGAP_ERR_INVALID, "umgap_synth_peptides_dev: null argument");
    int rc = use_device(device);
    if (rc) return rc;
    if (n) hipLaunchKernelGGL(k_synth_peptides, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, first, n, seed, d_offsets, d_bytes);
    UMGAP_HIP(hipGetLastError());
    return UMGAP_OK;
}

int umgap_synth_tryptic_dev(const uint64_t* d_ids, uint64_t first, size_t n, uint64_t seed, const uint64_t* d_offsets, uint8_t* d_bytes, int device, void* stream) {
    if (n && (!d_offsets || !d_bytes)) return fail(UMGAP_ERR_INVALID, "umgap_synth_tryptic_dev: null argument");
    int rc = use_device(device);
    if (rc) return rc;
    if (n) hipLaunchKernelGGL(k_synth_tryptic, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, d_ids, first, n, seed, d_offsets, d_bytes);
    UMGAP_HIP(hipGetLastError());
    return UMGAP_OK;
}

int umgap_synth_prot_reads_dev(const uint8_t* d_proteome, uint64_t n_proteome_fragments, uint32_t fragment_len, uint64_t seed, uint64_t first_read,
                               size_t n_reads, uint32_t fragments_per_read, uint32_t mutations_per_mille, uint8_t* d_out, int device, void* stream) {
    if (!d_proteome || !d_out || !n_proteome_fragments || !fragment_len || !fragments_per_read) return fail(UMGAP_ERR_INVALID, "umgap_synth_prot_reads_dev: bad argument");
    int rc = use_device(device);
    if (rc) return rc;
    if (n_reads) hipLaunchKernelGGL(k_synth_prot_reads, dim3(grid_for(n_reads * fragments_per_read * (size_t)fragment_len)), dim3(256), 0, (hipStream_t)stream,
                                    d_proteome, n_proteome_fragments, fragment_len, seed, first_read, n_reads, fragments_per_read, mutations_per_mille, d_out);
    UMGAP_HIP(hipGetLastError());
    return UMGAP_OK;
}

int umgap_synth_reads9_mix_dev(const umgap_taxonomy* tx, const uint32_t* d_taxon_of_row, const uint32_t* d_row_of_id, uint64_t n_tax,
                               uint64_t n_keys, uint64_t a, uint64_t b, uint64_t seed, uint64_t first_pair, uint64_t n_pairs,
                               uint32_t kmers_per_pair, uint32_t path_percent, uint32_t unrelated_percent, uint64_t* d_keys45, void* stream) {
    if (!tx || !d_taxon_of_row || !d_row_of_id || !d_keys45 || !n_tax || !n_keys || !kmers_per_pair) return fail(UMGAP_ERR_INVALID, "umgap_synth_reads9_dev: bad argument");
    if (n_keys >= 512000000000ull) return fail(UMGAP_ERR_INVALID, "umgap_synth_reads9_dev: n_keys must be < 20^9");
    if (path_percent + unrelated_percent > 99) return fail(UMGAP_ERR_INVALID, "umgap_synth_reads9_mix_dev: path + unrelated must leave at least 1 % of misses");
    int rc = use_device(tx->device);
    if (rc) return rc;
    TaxView V = view_of(tx, 0);
    if (n_pairs) hipLaunchKernelGGL(k_synth_reads, dim3(grid_for(n_pairs * kmers_per_pair)), dim3(256), 0, (hipStream_t)stream, V, d_taxon_of_row, d_row_of_id, n_tax, n_keys, a, b % 512000000000ull, seed, first_pair, n_pairs, kmers_per_pair, path_percent, unrelated_percent, d_keys45);
    UMGAP_HIP(hipGetLastError());
    return UMGAP_OK;
}

int umgap_synth_reads9_dev(const umgap_taxonomy* tx, const uint32_t* d_taxon_of_row, const uint32_t* d_row_of_id, uint64_t n_tax,
                           uint64_t n_keys, uint64_t a, uint64_t b, uint64_t seed, uint64_t first_pair, uint64_t n_pairs,
                           uint32_t kmers_per_pair, uint64_t* d_keys45, void* stream) {
    return umgap_synth_reads9_mix_dev(tx, d_taxon_of_row, d_row_of_id, n_tax, n_keys, a, b, seed, first_pair, n_pairs, kmers_per_pair, 60, 10, d_keys45, stream);
}

}  // extern "C"
