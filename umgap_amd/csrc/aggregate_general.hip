// aggregate_general.hip — MRTL, tree-hybrid and rmq-hybrid aggregation on the GPU (gfx950).
//
//   rmq::rtl::RTLCalculator   src/rmq/rtl.rs:39-57    (-m rmq  -a mrtl ; scripts/umgap-analyse.sh:282,293,298)
//   tree::mix::MixCalculator  src/tree/mix.rs:43-64   (-m tree -a hybrid; the taxa2agg default, :288)
//   rmq::mix::MixCalculator   src/rmq/mix.rs:57-93    (-m rmq  -a hybrid)
//
// One wavefront owns one read.  Per read, in the wave's private LDS scratch:
//   1. distinct taxa + multiplicities through an open-addressing hash table (ds_cmpst / ds_add);
//      scored reads (-s) elect leaders the same way but add the scores in input order
//      (agg::count, src/agg/mod.rs:32-34);
//   2. agg::filter (:39-44), then {pre,last} of the survivors; unknown ids end the read with the
//      reference's error;  3. survivors sorted by preorder index (rank by counting);
//   4. the strategy on the sorted list (closed forms: SURVEY.md §8a rows 16, 17, 20):
//      MRTL    score(x) = count(x) + sum of counts of its ancestors in the list, nearest first
//              (the order rtl.rs adds them); argmax.
//      hybrid  the induced ("virtual") tree is implicit in the sorted list plus the depth of the
//              LCA of each adjacent pair: a range [i,j] of the list is one subtree; its branching
//              node sits at m = min adjacent-LCA depth; elements with depth == m are chain hits
//              (Tree::collapse, src/tree/mod.rs:71-86); cuts at adjacent-LCA depth == m split it
//              into the children, whose values are plain range sums (Tree::aggregate, :90-101).
//      rmq mix candidates = the hits and the LCAs of adjacent hits (the LCA closure of the set);
//              lca / rtl weights by interval tests; argmax of lca*f + rtl*(1-f) in f32.
//   Ties (HashMap / HashSet order in the reference) -> smallest preorder index.
// Reads with more than CAP distinct taxa are deferred to a second launch of the same kernel with a
// large scratch (1 wave per workgroup); beyond that the read is flagged UMGAP_READ_UNSUPPORTED.
#include "agg_general.hpp"
#include "bound.hpp"

using namespace umgap;
using namespace umgap::agg;



namespace {

// the reads on the defer list: interval codes -> taxon ids, in place (one wave per read)
__global__ __launch_bounds__(256) void k_deferred_codes_to_ids(const uint32_t* __restrict__ todo, const uint32_t* __restrict__ todo_count, const uint64_t* __restrict__ offsets,
                                                               uint32_t* __restrict__ hits, const uint32_t* __restrict__ id_of_pre) {
    const uint32_t lane = threadIdx.x & 63;
    const size_t wave0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
    const size_t n = (size_t)*todo_count;
    for (size_t w = wave0; w < n; w += nwaves) {
        const size_t r = todo[w];
        for (uint64_t i = offsets[r] + lane; i < offsets[r + 1]; i += 64) {
            const uint32_t v = hits[i];
            if (v) hits[i] = id_of_pre[umgap::bound::pre_of(v)];
        }
    }
}

}  // namespace
namespace umgap {
void launch_deferred_codes_to_ids(const uint32_t* d_list, const uint32_t* d_count, const uint64_t* d_offsets, uint32_t* codes, const uint32_t* d_id_of_pre, hipStream_t stream) {
    hipLaunchKernelGGL(k_deferred_codes_to_ids, dim3(64), dim3(256), 0, stream, d_list, d_count, d_offsets, codes, d_id_of_pre);
}
}  // namespace umgap
namespace {

template <bool SCORED, int CAP, int HCAP, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_agg_general(TaxView V, int method, int strategy, float factor,
                                                            const uint32_t* __restrict__ taxa, const float* __restrict__ scores,
                                                            const uint64_t* __restrict__ offsets, uint32_t stride, size_t n_reads,
                                                            float lower_bound, uint32_t* __restrict__ out_taxon,
                                                            uint32_t* __restrict__ out_status,
                                                            const uint32_t* __restrict__ todo, const uint32_t* __restrict__ todo_count,
                                                            uint32_t* __restrict__ defer, uint32_t* __restrict__ defer_count) {
    __shared__ Scratch<CAP, HCAP> scratch[WAVES];
    Scratch<CAP, HCAP>& S = scratch[threadIdx.x >> 6];
    const uint32_t lane = threadIdx.x & 63;
    const size_t wave0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (uint32_t i = lane; i < HCAP; i += 64) { S.hkeys[i] = 0u; S.hcnt[i] = 0u; }
    wave_sync();
    const size_t n_work = todo ? (size_t)*todo_count : n_reads;

    for (size_t w = wave0; w < n_work; w += nwaves) {
        const size_t r = todo ? (size_t)todo[w] : w;
        uint64_t off, end;
        if (offsets) { off = offsets[r]; end = offsets[r + 1]; } else { off = (uint64_t)r * stride; end = off + stride; }
        uint32_t off_lo = __builtin_amdgcn_readfirstlane((uint32_t)off), off_hi = __builtin_amdgcn_readfirstlane((uint32_t)(off >> 32));
        off = ((uint64_t)off_hi << 32) | off_lo;
        uint32_t n_lo = __builtin_amdgcn_readfirstlane((uint32_t)(end - off)), n_hi = __builtin_amdgcn_readfirstlane((uint32_t)((end - off) >> 32));
        const uint64_t n = ((uint64_t)n_hi << 32) | n_lo;
        const uint32_t* rp = taxa + off;
        const float* sp = SCORED ? scores + off : nullptr;
        uint32_t res_taxon = 0, res_status = UMGAP_READ_OK;
        const bool ok = general_read<SCORED, CAP, HCAP>(V, S, method, strategy, factor, lower_bound, rp, sp, n, lane, res_taxon, res_status);
        if (!ok) {
            if (lane == 0) {
                if (defer) defer[atomicAdd(defer_count, 1u)] = (uint32_t)r;
                else { out_taxon[r] = 0; out_status[r] = UMGAP_READ_UNSUPPORTED; }
            }
            continue;
        }
        wave_sync();
        if (lane == 0) {
            out_taxon[r] = res_taxon;
            out_status[r] = res_status;
        }
    }
}

// Third pass: reads with more distinct taxa than the large LDS scratch holds (> 1024), one wave per read with its working
// set in global memory (ScratchGlobal).  Slow (the sort and the scans are quadratic in the distinct taxa) and rare.
template <bool SCORED>
__global__ __launch_bounds__(64) void k_agg_general_big(TaxView V, int method, int strategy, float factor, const uint32_t* __restrict__ taxa,
                                                        const float* __restrict__ scores, const uint64_t* __restrict__ offsets, uint32_t stride,
                                                        float lower_bound, uint32_t* __restrict__ out_taxon, uint32_t* __restrict__ out_status,
                                                        const uint32_t* __restrict__ todo, const uint32_t* __restrict__ todo_count,
                                                        uint32_t* __restrict__ scratch, uint32_t cap, uint32_t hcap) {
    ScratchGlobal S;
    S.carve(scratch + (size_t)blockIdx.x * ScratchGlobal::words(cap, hcap), cap, hcap);
    const uint32_t lane = threadIdx.x & 63;
    const size_t n_work = (size_t)*todo_count;
    for (size_t w = blockIdx.x; w < n_work; w += gridDim.x) {
        const size_t r = (size_t)todo[w];
        uint64_t off, end;
        if (offsets) { off = offsets[r]; end = offsets[r + 1]; } else { off = (uint64_t)r * stride; end = off + stride; }
        uint32_t res_taxon = 0, res_status = UMGAP_READ_OK;
        const bool ok = general_read_t<SCORED, ScratchGlobal>(V, S, method, strategy, factor, lower_bound, taxa + off, SCORED ? scores + off : nullptr,
                                                             end - off, lane, res_taxon, res_status);
        ScratchGlobal::sync();
        if (lane == 0) {
            out_taxon[r] = ok ? res_taxon : 0u;
            out_status[r] = ok ? res_status : (uint32_t)UMGAP_READ_UNSUPPORTED;
        }
    }
}

}  // namespace

namespace umgap {

int launch_strided_group(const TaxView& V, int method, int strategy, float factor, float lower_bound, const uint32_t* d_taxa, uint32_t stride,
                         size_t n_reads, uint32_t* d_out_taxon, uint32_t* d_out_status, uint32_t* d_list, uint32_t* d_count, hipStream_t stream);
int launch_csr_group(const TaxView& V, int method, int strategy, float factor, float lower_bound, const uint32_t* d_taxa, const uint64_t* d_offsets,
                     size_t n_reads, uint32_t* d_out_taxon, uint32_t* d_out_status, uint32_t* d_list, uint32_t* d_count, hipStream_t stream, bool bound_hits);

// Launches the general aggregation: the compact-scratch pass over all reads, the large-scratch pass over whatever it
// deferred (> 64 distinct taxa), and the global-memory pass over what that one deferred (> 1024).  `d_defer` must hold
// 2 * (n_reads + 1) u32; d_big = BIG_SLOTS read scratches of ScratchGlobal::words(BIG_CAP, BIG_HCAP) words, hash parts zero.
// bound_codes (non-null: the hits again, writable): they are interval codes of a table bound to the taxonomy.  The group pass works
// on the codes; the reads it defers (more than 64 distinct taxa or 128 hits -- a tryptic read has a handful) get their codes turned
// back into taxon ids in place before the large-scratch passes, which know ids only.
int launch_general(const TaxView& V, int method, int strategy, float factor, float lower_bound, const uint32_t* d_taxa,
                   const float* d_scores, const uint64_t* d_offsets, uint32_t stride, size_t n_reads, uint32_t* d_out_taxon,
                   uint32_t* d_out_status, uint32_t* d_defer, uint32_t* d_big, hipStream_t stream, uint32_t* bound_codes, const uint32_t* d_id_of_pre) {
    if (bound_codes && (d_scores || !d_offsets)) return fail(UMGAP_ERR_INVALID, "launch_general: bound hits come as unscored CSR reads");
    uint32_t* d_count = d_defer;      // [0] = number of deferred reads
    uint32_t* d_list = d_defer + 1;
    uint32_t* d_count2 = d_defer + n_reads + 1;
    uint32_t* d_list2 = d_count2 + 1;
    UMGAP_HIP(hipMemsetAsync(d_count, 0, sizeof(uint32_t), stream));
    UMGAP_HIP(hipMemsetAsync(d_count2, 0, sizeof(uint32_t), stream));
    size_t g = (n_reads + 3) / 4;
    if (g > 256 * 8) g = 256 * 8;
    if (g < 1) g = 1;
    dim3 grid((unsigned)g), block(256), big_grid(256), big_block(64);
    const bool grouped = !d_scores && !d_offsets && stride >= 1 && stride <= 128;   // aggregate_group.hip takes the first pass
    if (grouped) {
        int rc = launch_strided_group(V, method, strategy, factor, lower_bound, d_taxa, stride, n_reads, d_out_taxon, d_out_status, d_list, d_count, stream);
        if (rc) return rc;
        hipLaunchKernelGGL((k_agg_general<false, 1024, 4096, 1>), big_grid, big_block, 0, stream, V, method, strategy, factor, d_taxa, d_scores, d_offsets,
                           stride, n_reads, lower_bound, d_out_taxon, d_out_status, (const uint32_t*)d_list, (const uint32_t*)d_count, d_list2, d_count2);
        hipLaunchKernelGGL(k_agg_general_big<false>, dim3(BIG_SLOTS), dim3(64), 0, stream, V, method, strategy, factor, d_taxa, d_scores, d_offsets, stride,
                           lower_bound, d_out_taxon, d_out_status, (const uint32_t*)d_list2, (const uint32_t*)d_count2, d_big, BIG_CAP, BIG_HCAP);
    } else if (d_scores) {
        hipLaunchKernelGGL((k_agg_general<true, 64, 512, 4>), grid, block, 0, stream, V, method, strategy, factor, d_taxa, d_scores, d_offsets, stride,
                           n_reads, lower_bound, d_out_taxon, d_out_status, (const uint32_t*)nullptr, (const uint32_t*)nullptr, d_list, d_count);
        hipLaunchKernelGGL((k_agg_general<true, 1024, 4096, 1>), big_grid, big_block, 0, stream, V, method, strategy, factor, d_taxa, d_scores, d_offsets,
                           stride, n_reads, lower_bound, d_out_taxon, d_out_status, (const uint32_t*)d_list, (const uint32_t*)d_count, d_list2, d_count2);
        hipLaunchKernelGGL(k_agg_general_big<true>, dim3(BIG_SLOTS), dim3(64), 0, stream, V, method, strategy, factor, d_taxa, d_scores, d_offsets, stride,
                           lower_bound, d_out_taxon, d_out_status, (const uint32_t*)d_list2, (const uint32_t*)d_count2, d_big, BIG_CAP, BIG_HCAP);
    } else if (d_offsets) {
        // unscored CSR: the first pass batched over groups of reads (aggregate_group.hip), the rest as above
        int rc = launch_csr_group(V, method, strategy, factor, lower_bound, d_taxa, d_offsets, n_reads, d_out_taxon, d_out_status, d_list, d_count, stream, bound_codes != nullptr);
        if (rc) return rc;
        if (bound_codes) hipLaunchKernelGGL(k_deferred_codes_to_ids, dim3(64), dim3(256), 0, stream, (const uint32_t*)d_list, (const uint32_t*)d_count, d_offsets, bound_codes, d_id_of_pre);
        hipLaunchKernelGGL((k_agg_general<false, 1024, 4096, 1>), big_grid, big_block, 0, stream, V, method, strategy, factor, d_taxa, d_scores, d_offsets,
                           stride, n_reads, lower_bound, d_out_taxon, d_out_status, (const uint32_t*)d_list, (const uint32_t*)d_count, d_list2, d_count2);
        hipLaunchKernelGGL(k_agg_general_big<false>, dim3(BIG_SLOTS), dim3(64), 0, stream, V, method, strategy, factor, d_taxa, d_scores, d_offsets, stride,
                           lower_bound, d_out_taxon, d_out_status, (const uint32_t*)d_list2, (const uint32_t*)d_count2, d_big, BIG_CAP, BIG_HCAP);
    } else {
        hipLaunchKernelGGL((k_agg_general<false, 64, 256, 4>), grid, block, 0, stream, V, method, strategy, factor, d_taxa, d_scores, d_offsets, stride,
                           n_reads, lower_bound, d_out_taxon, d_out_status, (const uint32_t*)nullptr, (const uint32_t*)nullptr, d_list, d_count);
        hipLaunchKernelGGL((k_agg_general<false, 1024, 4096, 1>), big_grid, big_block, 0, stream, V, method, strategy, factor, d_taxa, d_scores, d_offsets,
                           stride, n_reads, lower_bound, d_out_taxon, d_out_status, (const uint32_t*)d_list, (const uint32_t*)d_count, d_list2, d_count2);
        hipLaunchKernelGGL(k_agg_general_big<false>, dim3(BIG_SLOTS), dim3(64), 0, stream, V, method, strategy, factor, d_taxa, d_scores, d_offsets, stride,
                           lower_bound, d_out_taxon, d_out_status, (const uint32_t*)d_list2, (const uint32_t*)d_count2, d_big, BIG_CAP, BIG_HCAP);
    }
    UMGAP_HIP(hipGetLastError());
    return UMGAP_OK;
}

}  // namespace umgap
