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

}  // namespace

namespace umgap {

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
