// prot_limits.hpp -- what one read may be for the fused surface-B kernels (k_classify_prot9, classify.hip): reads beyond these come
// back UMGAP_READ_UNSUPPORTED; the record path (records.hip) takes them through its general kernels instead.
#pragma once
#include <cstddef>
#include <cstdint>

namespace umgap {
constexpr uint32_t PROT_MAX_FRAGS = 16;     // fragments (FASTA records) per read
constexpr uint32_t PROT_MAX_BYTES = 384;    // residues per read, all fragments together
constexpr uint32_t PROT_MAX_WINDOWS = 128;  // 9-mer windows per read
// The LONG variant of the same kernels (round 4): a pair translated in all six frames (`translate -a`, scripts/umgap-analyse.sh:277-288:
// 12 fragments of ~50 residues, ~500 windows) -- two reads per wave instead of eight; after seedextend a read's hits are compacted,
// and the aggregate stage takes it when at most PROT_MAX_WINDOWS of them are left (a read's hits sit in the one or two frames that
// code; more than that and the read comes back UMGAP_READ_UNSUPPORTED like an over-long one).
constexpr uint32_t PROT_LONG_FRAGS = 32;
constexpr uint32_t PROT_LONG_BYTES = 1536;
constexpr uint32_t PROT_LONG_WINDOWS = 512;
// umgap_classify_prot9[_strategy]_dev picks the LONG variant for reads of more than PROT_SHORT_FRAGS fragments
constexpr uint32_t PROT_SHORT_FRAGS = 4;
}  // namespace umgap

struct umgap_index;
struct umgap_taxonomy;
namespace umgap {
// umgap_classify_prot9_strategy_dev with a gate (classify.hip): a device pointer to three words {offsets bad, a read does not fit the
// short variant, a read does not fit the LONG variant}, nullable.  variant 0 / 1 = short / LONG without a gate (what the device call
// picks from fragments_per_read); with a gate the caller launches BOTH and the device decides which one works: the short one runs
// when no word is set, the LONG one when only word 1 is (records.hip: no host round trip between the check kernel and the launch).
int classify_prot9_gated(umgap_index* ix, umgap_taxonomy* tx, int method, int strategy, float factor, float lower_bound, int ranked_only,
                         const uint8_t* d_bytes, const uint64_t* d_frag_offsets, size_t n_reads, uint32_t fragments_per_read, int seedextend,
                         uint32_t min_seed_size, uint32_t max_gap_size, uint32_t* d_out_taxon, uint32_t* d_out_status, const uint32_t* gate, int long_variant,
                         void* stream);
}  // namespace umgap
