// records.hip — the 9-mer pipeline over a block of FASTA records held in host memory:
//   prot2kmer2lca -o <index> | seedextend -s S -g G | uniq [-d D] | taxa2agg <options> <taxons>
// (scripts/umgap-analyse.sh:257-310) as ONE C-ABI call.  The text between the four reference
// commands (84 decimal taxon ids per read pair, three pipes) never exists: the host hands over the
// protein fragments (what prot2kmer2lca reads, src/commands/prot2kmer2lca.rs:172-179) and which
// consecutive fragments `uniq` would merge (src/commands/uniq.rs:66-79); the per-window hits stay in
// HBM between the probe, seedextend (src/commands/seedextend.rs:99-148, in place) and the per-read
// aggregate (src/commands/taxa2agg.rs:160-180).  Only one taxon + status per read travels back.
#include "common.hpp"
#include "taxonomy.hpp"

using namespace umgap;

extern "C" int umgap_classify_records(umgap_index* ix, umgap_taxonomy* tx, int method, int strategy, float factor, float lower_bound,
                                      int ranked_only, const uint8_t* bytes, const uint64_t* frag_offsets, size_t n_fragments,
                                      const uint64_t* read_frags, size_t n_reads, int seedextend, uint32_t min_seed_size,
                                      uint32_t max_gap_size, uint32_t* out_taxon, uint32_t* out_status) {
    if (!ix || !tx || (n_reads && (!read_frags || !out_taxon || !out_status)) || (n_fragments && !frag_offsets))
        return fail(UMGAP_ERR_INVALID, "umgap_classify_records: null argument");
    if (ix->bound_to) return fail(UMGAP_ERR_INVALID, "the index is bound to a taxonomy (umgap_index_bind_taxonomy): its values are interval codes, not taxon ids; unbind it first");
    if (ix->kind != 9) return fail(UMGAP_ERR_INVALID, "umgap_classify_records: not a 9-mer index");
    if (ix->device != tx->device) return fail(UMGAP_ERR_INVALID, "umgap_classify_records: index and taxonomy live on different devices");
    if (!n_reads) return UMGAP_OK;
    if (read_frags[0] != 0 || read_frags[n_reads] != n_fragments) return fail(UMGAP_ERR_INVALID, "umgap_classify_records: read_frags must cover [0, n_fragments]");
    int rc = use_device(tx->device);
    if (rc) return rc;
    const uint64_t base = n_fragments ? frag_offsets[0] : 0, n_bytes = n_fragments ? frag_offsets[n_fragments] - base : 0;
    if (n_bytes && !bytes) return fail(UMGAP_ERR_INVALID, "umgap_classify_records: null bytes");
    // window CSR: fragment f owns hits [win[f], win[f+1]), read r owns [rwin[r], rwin[r+1])
    std::vector<uint64_t> rel(n_fragments + 1), win(n_fragments + 1), rwin(n_reads + 1);
    win[0] = 0;
    rel[0] = 0;
    for (size_t f = 0; f < n_fragments; f++) {
        if (frag_offsets[f + 1] < frag_offsets[f]) return fail(UMGAP_ERR_INVALID, "umgap_classify_records: frag_offsets must not decrease");
        uint64_t len = frag_offsets[f + 1] - frag_offsets[f];
        rel[f + 1] = frag_offsets[f + 1] - base;
        win[f + 1] = win[f] + (len >= 9 ? len - 8 : 0);
    }
    for (size_t r = 0; r <= n_reads; r++) {
        if (read_frags[r] > n_fragments || (r && read_frags[r] < read_frags[r - 1])) return fail(UMGAP_ERR_INVALID, "umgap_classify_records: bad read_frags");
        rwin[r] = win[read_frags[r]];
    }
    const uint64_t n_win = win[n_fragments];
    auto pad = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t b_bytes = pad(n_bytes + 16), b_off = pad((n_fragments + 1) * 8), b_roff = pad((n_reads + 1) * 8), b_hits = pad(n_win * 4 + 16),
                 b_out = pad(n_reads * 4);
    const size_t need = b_bytes + 2 * b_off + b_roff + b_hits + 2 * b_out;
    if (need > tx->d_stage_bytes) {
        if (tx->d_stage) (void)hipFree(tx->d_stage);
        tx->d_stage = nullptr;
        tx->d_stage_bytes = 0;
        hipError_t e = hipMalloc(&tx->d_stage, need + need / 4);
        if (e != hipSuccess) return fail(UMGAP_ERR_NOMEM, std::string("hipMalloc(record staging): ") + hipGetErrorString(e));
        tx->d_stage_bytes = need + need / 4;
    }
    char* p = (char*)tx->d_stage;
    uint8_t* d_bytes = (uint8_t*)p; p += b_bytes;
    uint64_t* d_off = (uint64_t*)p; p += b_off;
    uint64_t* d_win = (uint64_t*)p; p += b_off;
    uint64_t* d_rwin = (uint64_t*)p; p += b_roff;
    uint32_t* d_hits = (uint32_t*)p; p += b_hits;
    uint32_t* d_ot = (uint32_t*)p; p += b_out;
    uint32_t* d_os = (uint32_t*)p;
    hipStream_t s = tx->stream;
    if (n_bytes) UMGAP_HIP(hipMemcpyAsync(d_bytes, bytes + base, n_bytes, hipMemcpyHostToDevice, s));
    UMGAP_HIP(hipMemcpyAsync(d_off, rel.data(), (n_fragments + 1) * 8, hipMemcpyHostToDevice, s));
    UMGAP_HIP(hipMemcpyAsync(d_win, win.data(), (n_fragments + 1) * 8, hipMemcpyHostToDevice, s));
    UMGAP_HIP(hipMemcpyAsync(d_rwin, rwin.data(), (n_reads + 1) * 8, hipMemcpyHostToDevice, s));
    if (n_win) {
        rc = umgap_index_lookup_prot9_dev(ix, d_bytes, d_off, d_win, n_fragments, 0u, d_hits, s);
        if (!rc && seedextend) rc = umgap_seedextend_dev(d_win, n_fragments, min_seed_size, max_gap_size, d_hits, nullptr, tx->device, s);
    }
    if (!rc) rc = umgap_aggregate_dev(tx, method, strategy, factor, lower_bound, ranked_only, d_hits, nullptr, d_rwin, n_reads, d_ot, d_os, s);
    if (rc) { (void)hipStreamSynchronize(s); return rc; }
    UMGAP_HIP(hipMemcpyAsync(out_taxon, d_ot, n_reads * 4, hipMemcpyDeviceToHost, s));
    UMGAP_HIP(hipMemcpyAsync(out_status, d_os, n_reads * 4, hipMemcpyDeviceToHost, s));
    UMGAP_HIP(hipStreamSynchronize(s));
    return UMGAP_OK;
}
