// prot_limits.hpp -- what one read may be for the fused surface-B kernels (k_classify_prot9, classify.hip): reads beyond these come
// back UMGAP_READ_UNSUPPORTED; the record path (records.hip) takes them through its general kernels instead.
#pragma once
#include <cstddef>
#include <cstdint>

namespace umgap {
constexpr uint32_t PROT_MAX_FRAGS = 16;     // fragments (FASTA records) per read
constexpr uint32_t PROT_MAX_BYTES = 384;    // residues per read, all fragments together
constexpr uint32_t PROT_MAX_WINDOWS = 128;  // 9-mer windows per read
}  // namespace umgap

struct umgap_index;
struct umgap_taxonomy;
namespace umgap {
// umgap_classify_prot9_strategy_dev with a gate (classify.hip): a device pointer to two words, nullable; the launch does nothing
// when either is set by the time it runs
int classify_prot9_gated(umgap_index* ix, umgap_taxonomy* tx, int method, int strategy, float factor, float lower_bound, int ranked_only,
                         const uint8_t* d_bytes, const uint64_t* d_frag_offsets, size_t n_reads, uint32_t fragments_per_read, int seedextend,
                         uint32_t min_seed_size, uint32_t max_gap_size, uint32_t* d_out_taxon, uint32_t* d_out_status, const uint32_t* gate, void* stream);
}  // namespace umgap
