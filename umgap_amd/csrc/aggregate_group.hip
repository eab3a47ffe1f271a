// aggregate_group.hip — umgap_aggregate_strided_dev for unscored reads of at most 128 taxa: a wave takes a GROUP of reads
// (as many as fit its 1024-word LDS strip), copies their taxa from HBM into the strip with coalesced loads and aggregates
// them together (agg_group.hpp: count / filter per read in LDS, the taxonomy round trips once per group), instead of one
// read per wave with 2-4 dependent round trips each (aggregate.hip / aggregate_general.hip, which keep the scored reads, the
// CSR layout and the long reads).  The tryptic pipelines' reads are short -- 16 hits fill a quarter of a wave -- so this is
// where the per-read kernels wasted most (BASELINE configs[4]: MRTL 1.2 -> see profiles/r02_config5_tryptic.json).
// Reference: src/commands/taxa2agg.rs:161-180 with the aggregators of src/tree/{lca,mix}.rs, src/rmq/{lca,rtl,mix}.rs.
#include "agg_group.hpp"

using namespace umgap;
using namespace umgap::agg;

namespace {

constexpr uint32_t STRIP = 1024;

__global__ __launch_bounds__(256) void k_agg_strided_group(TaxView V, int method, int strategy, float factor, const uint32_t* __restrict__ taxa,
                                                           uint32_t stride, size_t n_reads, float lower_bound, uint32_t* __restrict__ out_taxon,
                                                           uint32_t* __restrict__ out_status, uint32_t* __restrict__ defer, uint32_t* __restrict__ defer_count) {
    __shared__ uint32_t lds_taxa[4][STRIP];
    __shared__ GroupScratch scratch[4];
    const uint32_t wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint32_t* wtaxa = lds_taxa[wv];
    GroupScratch& S = scratch[wv];
    for (uint32_t i = lane; i < 256; i += 64) { S.hkeys[i] = 0u; S.hcnt[i] = 0u; }
    wave_sync();
    const uint32_t G = STRIP / stride;
    const size_t wave0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
    const size_t n_groups = (n_reads + G - 1) / G;
    for (size_t g = wave0; g < n_groups; g += nwaves) {
        const size_t r0 = g * G;
        const uint32_t nr = (uint32_t)min((size_t)G, n_reads - r0), nk = nr * stride;
        const uint32_t* src = taxa + r0 * stride;
        for (uint32_t i = lane; i < nk; i += 64) wtaxa[i] = __builtin_nontemporal_load(src + i);
        wave_sync();
        general_group<false>(V, S, method, strategy, factor, lower_bound, wtaxa, stride, nr, lane, 0ull, 0u, r0, out_taxon, out_status, defer, defer_count);
        wave_sync();
    }
}

// The same for the CSR layout (umgap_aggregate_dev: ragged reads, what taxa2agg and the tryptic pipeline produce).  A wave owns a
// CHUNK of 64 consecutive reads (lane k: the span of read k) and works through it in sub-rounds: the stride of a sub-round is the
// power of two that holds the longest read still to do (at most 128; longer reads go to the defer list), as many reads as then fit
// the strip are laid out at that stride -- every strip slot fetches its own taxon, the span coming from the owning lane by
// ds_bpermute -- and aggregated together.  Short reads (tryptic: 4-20 hits) run 32-64 to a sub-round instead of one per wave.
// BOUND: the hits are interval codes of a table bound to the taxonomy (bound.hpp) -- no by-id gather per distinct taxon.
// MODE (agg_group.hpp): one instantiation per strategy -- MRTL, tree hybrid, rmq hybrid -- so that each carries its own code and
// registers only: one kernel for all three took 144 registers = 3 waves per SIMD; now 98 / 128 (capped) / 111 = 4, which is also what
// its 40 KB of LDS per workgroup allow.  The kernel is bound by the latency of its 2-6 dependent taxonomy round trips per sub-batch:
// 3 -> 4 waves per SIMD took MRTL on 10 M four-hit reads from 506 to 375 us and the tree hybrid from 1273 to 884 (round 5).  A fifth
// wave (a 512-word strip for short reads: 32 KB per workgroup, MRTL at 96 registers) measured the same on the same box, so the strip
// stays one size.
template <bool BOUND, int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_agg_csr_group(TaxView V, int method, int strategy, float factor, const uint32_t* __restrict__ taxa,
                                                       const uint64_t* __restrict__ offsets, size_t n_reads, float lower_bound, uint32_t* __restrict__ out_taxon,
                                                       uint32_t* __restrict__ out_status, uint32_t* __restrict__ defer, uint32_t* __restrict__ defer_count) {
    __shared__ uint32_t lds_taxa[4][STRIP];
    __shared__ GroupScratch scratch[4];
    const uint32_t wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint32_t* wtaxa = lds_taxa[wv];
    GroupScratch& S = scratch[wv];
    for (uint32_t i = lane; i < 256; i += 64) { S.hkeys[i] = 0u; S.hcnt[i] = 0u; }
    wave_sync();
    const size_t wave0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
    const size_t n_chunks = (n_reads + 63) / 64;
    for (size_t c = wave0; c < n_chunks; c += nwaves) {
        const size_t r0 = c * 64;
        const uint32_t nr_chunk = (uint32_t)min((size_t)64, n_reads - r0);
        // lane k: where read r0 + k lies (index clamped: lanes past the end see an empty read)
        const uint64_t o = offsets[min(r0 + lane, n_reads)], o1 = offsets[min(r0 + lane + 1, n_reads)];
        const uint64_t l64 = o1 >= o ? o1 - o : 0ull;
        const bool big = l64 > 128;                                        // does not fit a strip row: the large-scratch pass takes it
        const uint32_t len = big ? 0u : (uint32_t)l64;
        const uint32_t o_lo = (uint32_t)o, o_hi = (uint32_t)(o >> 32);
        uint32_t k0 = 0;
        while (k0 < nr_chunk) {
            // stride: the power of two that holds the longest read from k0 on
            uint32_t m = lane >= k0 && lane < nr_chunk ? len : 0u;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) { const uint32_t t = (uint32_t)__shfl_xor((int)m, d, 64); m = t > m ? t : m; }
            uint32_t stride = 1;
            while (stride < m) stride <<= 1;                                // wave-uniform
            const uint32_t sh = (uint32_t)__builtin_ctz(stride);
            const uint32_t nr = min(nr_chunk - k0, STRIP >> sh);
            const uint32_t nk = nr << sh;
            for (uint32_t i = lane; i < ((nk + 63u) & ~63u); i += 64) {     // every lane takes part in the shuffles
                const uint32_t k = i >> sh, j = i & (stride - 1u);
                const int src = (int)((k0 + k) & 63u);
                const uint32_t lk = (uint32_t)__shfl((int)len, src, 64);
                const uint64_t ok = (uint64_t)(uint32_t)__shfl((int)o_lo, src, 64) | ((uint64_t)(uint32_t)__shfl((int)o_hi, src, 64) << 32);
                if (i < nk) wtaxa[i] = j < lk ? __builtin_nontemporal_load(taxa + ok + j) : 0u;
            }
            wave_sync();
            const uint64_t too_long = (__ballot(big) >> k0) & (nr >= 64 ? ~0ull : ((1ull << nr) - 1ull));
            general_group<BOUND, MODE>(V, S, method, strategy, factor, lower_bound, wtaxa, stride, nr, lane, 0ull, 0u, r0 + k0, out_taxon, out_status, defer, defer_count, 0ull,
                                       too_long);
            wave_sync();
            k0 += nr;
        }
    }
}

// LCA* over CSR reads the same way (round 5): the sub-rounds of k_agg_csr_group, the group's reads through lca_group (agg_group.hpp:
// count / filter per read in LDS, {pre, last} of all survivors in one round trip, the range minimum and the snap one lane per read).
// Reads of more than 128 hits do not fit a strip row: they are answered "empty" here and put on `todo` for the per-read kernel
// (aggregate.hip k_agg_lca), which runs behind this one over that list and overwrites them.
template <bool BOUND>
__global__ __launch_bounds__(256) void k_agg_csr_lca_group(TaxView V, int method, const uint32_t* __restrict__ taxa, const uint64_t* __restrict__ offsets, size_t n_reads,
                                                           float lower_bound, uint32_t* __restrict__ out_taxon, uint32_t* __restrict__ out_status,
                                                           uint32_t* __restrict__ todo, uint32_t* __restrict__ todo_count) {
    __shared__ uint32_t lds_taxa[4][STRIP];
    __shared__ LcaScratch scratch[4];
    const uint32_t wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint32_t* wtaxa = lds_taxa[wv];
    LcaScratch& S = scratch[wv];
    const bool tree_like = method != UMGAP_METHOD_RMQ;
    for (uint32_t i = lane; i < 256; i += 64) { S.hkeys[i] = 0u; S.hcnt[i] = 0u; }
    wave_sync();
    const size_t wave0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
    const size_t n_chunks = (n_reads + 63) / 64;
    for (size_t c = wave0; c < n_chunks; c += nwaves) {
        const size_t r0 = c * 64;
        const uint32_t nr_chunk = (uint32_t)min((size_t)64, n_reads - r0);
        const uint64_t o = offsets[min(r0 + lane, n_reads)], o1 = offsets[min(r0 + lane + 1, n_reads)];
        const uint64_t l64 = o1 >= o ? o1 - o : 0ull;
        const bool big = lane < nr_chunk && l64 > 128;
        if (big) todo[atomicAdd(todo_count, 1u)] = (uint32_t)(r0 + lane);
        const uint32_t len = big ? 0u : (uint32_t)l64;
        const uint32_t o_lo = (uint32_t)o, o_hi = (uint32_t)(o >> 32);
        uint32_t k0 = 0;
        while (k0 < nr_chunk) {
            uint32_t m = lane >= k0 && lane < nr_chunk ? len : 0u;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) { const uint32_t t = (uint32_t)__shfl_xor((int)m, d, 64); m = t > m ? t : m; }
            uint32_t stride = 1;
            while (stride < m) stride <<= 1;                                // wave-uniform
            const uint32_t sh = (uint32_t)__builtin_ctz(stride);
            const uint32_t nr = min(nr_chunk - k0, STRIP >> sh);
            const uint32_t nk = nr << sh;
            for (uint32_t i = lane; i < ((nk + 63u) & ~63u); i += 64) {     // every lane takes part in the shuffles
                const uint32_t k = i >> sh, j = i & (stride - 1u);
                const int src = (int)((k0 + k) & 63u);
                const uint32_t lk = (uint32_t)__shfl((int)len, src, 64);
                const uint64_t ok = (uint64_t)(uint32_t)__shfl((int)o_lo, src, 64) | ((uint64_t)(uint32_t)__shfl((int)o_hi, src, 64) << 32);
                if (i < nk) wtaxa[i] = j < lk ? __builtin_nontemporal_load(taxa + ok + j) : 0u;
            }
            wave_sync();
            lca_group<BOUND>(V, S, tree_like, lower_bound, wtaxa, stride, nr, lane, 0ull, 0u, r0 + k0, out_taxon, out_status,
                             [&](uint32_t k, uint32_t& res_taxon, uint32_t& res_status) {
                                 if (BOUND) lca_star_bound_from_lds(V, wtaxa + k * stride, stride, lower_bound, S.hkeys, S.hcnt, lane, res_taxon, res_status);
                                 else lca_star_from_lds(V, tree_like, wtaxa + k * stride, stride, lower_bound, S.hkeys, S.hcnt, lane, res_taxon, res_status);
                             });
            wave_sync();
            k0 += nr;
        }
    }
}

}  // namespace

namespace umgap {

int launch_csr_lca_group(const TaxView& V, int method, float lower_bound, const uint32_t* d_taxa, const uint64_t* d_offsets, size_t n_reads, uint32_t* d_out_taxon,
                         uint32_t* d_out_status, uint32_t* d_list, uint32_t* d_count, hipStream_t stream, bool bound_hits) {
    size_t g = ((n_reads + 63) / 64 + 3) / 4;
    if (g > 256 * 8) g = 256 * 8;
    if (g < 1) g = 1;
    if (bound_hits) hipLaunchKernelGGL(k_agg_csr_lca_group<true>, dim3((unsigned)g), dim3(256), 0, stream, V, method, d_taxa, d_offsets, n_reads, lower_bound, d_out_taxon,
                                       d_out_status, d_list, d_count);
    else hipLaunchKernelGGL(k_agg_csr_lca_group<false>, dim3((unsigned)g), dim3(256), 0, stream, V, method, d_taxa, d_offsets, n_reads, lower_bound, d_out_taxon, d_out_status,
                            d_list, d_count);
    UMGAP_HIP(hipGetLastError());
    return UMGAP_OK;
}

int launch_csr_group(const TaxView& V, int method, int strategy, float factor, float lower_bound, const uint32_t* d_taxa, const uint64_t* d_offsets,
                     size_t n_reads, uint32_t* d_out_taxon, uint32_t* d_out_status, uint32_t* d_list, uint32_t* d_count, hipStream_t stream, bool bound_hits) {
    size_t g = ((n_reads + 63) / 64 + 3) / 4;
    if (g > 256 * 8) g = 256 * 8;
    if (g < 1) g = 1;
    if (strategy == UMGAP_STRATEGY_LCA_STAR) return fail(UMGAP_ERR_INVALID, "launch_csr_group: LCA* has its own kernel");
    // one instantiation per strategy (see the kernel)
    const int mode = strategy == UMGAP_STRATEGY_MRTL ? 1 : (method == UMGAP_METHOD_TREE ? 2 : 3);
    auto go = [&](auto B, auto M) {
        hipLaunchKernelGGL((k_agg_csr_group<decltype(B)::value, decltype(M)::value>), dim3((unsigned)g), dim3(256), 0, stream, V, method, strategy, factor, d_taxa, d_offsets,
                           n_reads, lower_bound, d_out_taxon, d_out_status, d_list, d_count);
    };
    auto with_mode = [&](auto B) {
        if (mode == 1) go(B, std::integral_constant<int, 1>{});
        else if (mode == 2) go(B, std::integral_constant<int, 2>{});
        else go(B, std::integral_constant<int, 3>{});
    };
    if (bound_hits) with_mode(std::true_type{});
    else with_mode(std::false_type{});
    UMGAP_HIP(hipGetLastError());
    return UMGAP_OK;
}

// MRTL and the hybrids (LCA* keeps its per-read kernel in aggregate.hip, which measured 5 % faster on 16-hit reads): reads with
// more than 64 distinct taxa land in the defer list (d_list, *d_count, zeroed by the caller), which the caller's large-scratch
// passes work through.
int launch_strided_group(const TaxView& V, int method, int strategy, float factor, float lower_bound, const uint32_t* d_taxa, uint32_t stride,
                         size_t n_reads, uint32_t* d_out_taxon, uint32_t* d_out_status, uint32_t* d_list, uint32_t* d_count, hipStream_t stream) {
    const uint32_t G = STRIP / stride;
    size_t g = ((n_reads + G - 1) / G + 3) / 4;
    if (g > 256 * 8) g = 256 * 8;
    if (g < 1) g = 1;
    if (strategy == UMGAP_STRATEGY_LCA_STAR) return fail(UMGAP_ERR_INVALID, "launch_strided_group: LCA* has its own kernel");
    hipLaunchKernelGGL(k_agg_strided_group, dim3((unsigned)g), dim3(256), 0, stream, V, method, strategy, factor, d_taxa, stride, n_reads, lower_bound,
                       d_out_taxon, d_out_status, d_list, d_count);
    UMGAP_HIP(hipGetLastError());
    return UMGAP_OK;
}

}  // namespace umgap
