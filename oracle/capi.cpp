// capi.cpp — extern "C" surface of the CPU oracle for ctypes.  TEST INFRASTRUCTURE ONLY
// (see umgap_oracle.hpp header).  Nothing in the product links or loads this.
#include <atomic>
#include <thread>

#include "umgap_oracle.hpp"
#include "synth_oracle.hpp"

using namespace umgap_oracle;

namespace {
void set_err(char* err, size_t cap, const std::string& m) {
    if (!err || !cap) return;
    size_t n = std::min(cap - 1, m.size());
    memcpy(err, m.data(), n);
    err[n] = 0;
}
const uint64_t NONE = UINT64_MAX;
}  // namespace

extern "C" {

// ---- status codes shared with include/umgap_hip.h ---------------------------------------
enum { ST_OK = 0, ST_EMPTY = 1, ST_UNKNOWN_TAXON = 2, ST_PANIC = 3 };

// ---- parsing golden hooks -----------------------------------------------------------------
// returns -1 on success, else the ErrKind ordinal
int orc_parse_taxon(const char* line, size_t len, uint64_t* id, char* name, size_t name_cap, int* rank,
                    uint64_t* parent, int* valid) {
    try {
        Taxon t = parse_taxon(std::string(line, len));
        *id = t.id; *rank = t.rank; *parent = t.parent; *valid = t.valid;
        set_err(name, name_cap, t.name);
        return -1;
    } catch (Error& e) { return (int)e.kind; }
}
int orc_parse_usize(const char* s, size_t len, uint64_t* out) {
    try { *out = parse_usize(std::string(s, len)); return -1; } catch (Error& e) { return (int)e.kind; }
}
int orc_parse_f32(const char* s, size_t len, float* out) {
    try { *out = parse_f32(std::string(s, len)); return -1; } catch (Error& e) { return (int)e.kind; }
}
int64_t orc_rank_score(int rank) { auto s = rank_score(rank); return s ? (int64_t)*s : -1; }
const char* orc_rank_name(int rank) { return (rank >= 0 && rank < 32) ? RANK_NAMES[rank] : nullptr; }

// ---- taxonomy -----------------------------------------------------------------------------
// returns handle or NULL; *panicked = 1 when the reference would panic (exit 101)
void* orc_taxonomy_from_tsv(const char* path, char* err, size_t cap, int* panicked) {
    *panicked = 0;
    try { return new Taxonomy(read_taxa_file(path)); }
    catch (Error& e) { set_err(err, cap, e.msg); return nullptr; }
    catch (Panic& p) { *panicked = 1; set_err(err, cap, p.msg); return nullptr; }
}
void* orc_taxonomy_from_arrays(const uint64_t* id, const uint64_t* parent, const uint8_t* valid,
                               const uint8_t* rank, size_t n, char* err, size_t cap, int* panicked) {
    *panicked = 0;
    try {
        std::vector<Taxon> v;
        v.reserve(n);
        for (size_t i = 0; i < n; i++) v.push_back(Taxon{id[i], "", rank[i], parent[i], valid[i] != 0});
        return new Taxonomy(std::move(v));
    } catch (Error& e) { set_err(err, cap, e.msg); return nullptr; }
    catch (Panic& p) { *panicked = 1; set_err(err, cap, p.msg); return nullptr; }
}
void orc_taxonomy_free(void* h) { delete (Taxonomy*)h; }
uint64_t orc_taxonomy_root(void* h) { return ((Taxonomy*)h)->tree.root; }
size_t orc_taxonomy_len(void* h) { return ((Taxonomy*)h)->by_id.v.size(); }
size_t orc_taxonomy_count(void* h) { return ((Taxonomy*)h)->taxons.size(); }
void orc_taxonomy_rows(void* h, uint64_t* id, uint64_t* parent, uint8_t* valid, uint8_t* rank) {
    auto& t = ((Taxonomy*)h)->taxons;
    for (size_t i = 0; i < t.size(); i++) { id[i] = t[i].id; parent[i] = t[i].parent; valid[i] = t[i].valid; rank[i] = (uint8_t)t[i].rank; }
}
// out[len]; NONE where Option is None
void orc_ancestry(void* h, uint64_t* out) {
    auto a = ((Taxonomy*)h)->by_id.ancestry();
    for (size_t i = 0; i < a.size(); i++) out[i] = a[i] ? *a[i] : NONE;
}
int orc_snapping(void* h, int ranked, uint64_t* out) {
    try {
        auto s = ((Taxonomy*)h)->snapping(ranked != 0);
        for (size_t i = 0; i < s.size(); i++) out[i] = s[i] ? *s[i] : NONE;
        return 0;
    } catch (Panic&) { return ST_PANIC; }
}
size_t orc_euler(void* h, uint64_t* taxa, uint64_t* depth, size_t cap) {
    auto e = euler_tour(((Taxonomy*)h)->tree);
    for (size_t i = 0; i < e.size() && i < cap; i++) { taxa[i] = e[i].first; depth[i] = e[i].second; }
    return e.size();
}
// lineage: out[32] taxon id per rank or NONE; returns 0 / ST_UNKNOWN_TAXON (payload in *bad)
int orc_lineage(void* h, uint64_t index, uint64_t* out, uint64_t* bad) {
    try {
        auto l = ((Taxonomy*)h)->by_id.lineage(index);
        for (int i = 0; i < 32; i++) out[i] = l[i] ? l[i]->id : NONE;
        return 0;
    } catch (Error& e) { *bad = e.taxon; return ST_UNKNOWN_TAXON; }
}
int64_t orc_taxon_score(void* h, uint64_t index, int with_unknown) {
    Taxonomy* t = (Taxonomy*)h;
    TaxonList l = with_unknown ? TaxonList::new_with_unknown(t->taxons, true) : TaxonList(t->taxons);
    auto s = l.score(index);
    return s ? (int64_t)*s : -1;
}

// ---- RMQ golden hooks ---------------------------------------------------------------------
size_t orc_rmq_block_min(const uint64_t* a, size_t n, uint64_t* out) {
    std::vector<size_t> arr(a, a + n);
    auto b = RMQ::block_min(arr);
    for (size_t i = 0; i < b.size(); i++) out[i] = b[i];
    return b.size();
}
// flat output; lens[level] = length of that level; returns number of levels
size_t orc_rmq_sparse(const uint64_t* a, size_t n, const uint64_t* bm, size_t nb, uint64_t* out, uint64_t* lens,
                      size_t cap_levels) {
    std::vector<size_t> arr(a, a + n), b(bm, bm + nb);
    auto sp = RMQ::sparse(arr, b);
    size_t k = 0;
    for (size_t l = 0; l < sp.size() && l < cap_levels; l++) {
        lens[l] = sp[l].size();
        for (auto v : sp[l]) out[k++] = v;
    }
    return sp.size();
}
void orc_rmq_labels(const uint64_t* a, size_t n, uint64_t* out) {
    std::vector<size_t> arr(a, a + n);
    auto l = RMQ::labels(arr);
    for (size_t i = 0; i < n; i++) out[i] = l[i];
}
void* orc_rmq_new(const uint64_t* a, size_t n) { return new RMQ(std::vector<size_t>(a, a + n)); }
void orc_rmq_free(void* r) { delete (RMQ*)r; }
uint64_t orc_rmq_query(void* r, uint64_t s, uint64_t e) { return ((RMQ*)r)->query(s, e); }

// ---- aggregators ----------------------------------------------------------------------------
struct AggHandle {
    Taxonomy* tx;
    std::unique_ptr<Aggregator> agg;
    RmqLCACalculator* rmq_lca = nullptr;  // when method rmq + lca*, for orc_lca2
};
// method: 0 tree, 1 rmq; strategy: 0 lca*, 1 hybrid, 2 mrtl
void* orc_aggregator_new(void* tax, int method, int strategy, float factor, char* err, size_t cap) {
    try {
        auto* h = new AggHandle();
        h->tx = (Taxonomy*)tax;
        h->agg = make_aggregator(*h->tx, method ? Method::RangeMinimumQuery : Method::Tree,
                                 strategy == 0 ? Strategy::LowestCommonAncestor
                                               : (strategy == 1 ? Strategy::Hybrid : Strategy::MaximumRootToLeafPath),
                                 factor, nullptr);
        if (method == 1 && strategy == 0) h->rmq_lca = dynamic_cast<RmqLCACalculator*>(h->agg.get());
        return h;
    } catch (Error& e) { set_err(err, cap, e.msg); return nullptr; }
    catch (Panic& p) { set_err(err, cap, std::string("panic: ") + p.msg); return nullptr; }
}
void orc_aggregator_free(void* h) { delete (AggHandle*)h; }

// pairwise lca of rmq::lca (src/rmq/lca.rs:42-47); returns 0 or ST_UNKNOWN_TAXON
int orc_lca2(void* h, uint64_t a, uint64_t b, uint64_t* out) {
    AggHandle* ah = (AggHandle*)h;
    if (!ah->rmq_lca) return -1;
    try { *out = ah->rmq_lca->lca(a, b); return 0; } catch (Error& e) { *out = e.taxon; return ST_UNKNOWN_TAXON; }
}

// counting_aggregate / aggregate on one list; flags bit0 = EmptyInput, bit1 = may panic
int orc_aggregate(void* h, const uint64_t* taxa, const float* scores, size_t n, uint64_t* ok, size_t* n_ok,
                  size_t cap_ok, uint64_t* unk, size_t* n_unk, size_t cap_unk, int* flags) {
    AggHandle* ah = (AggHandle*)h;
    std::vector<std::pair<TaxonId, float>> v;
    for (size_t i = 0; i < n; i++) v.emplace_back(taxa[i], scores ? scores[i] : 1.0f);
    AggOutcome o;
    try { o = ah->agg->aggregate(count(v)); } catch (Panic&) { o.may_panic = true; }
    size_t k = 0;
    for (auto t : o.ok) if (k < cap_ok) ok[k++] = t;
    *n_ok = o.ok.size();
    k = 0;
    for (auto t : o.unknown) if (k < cap_unk) unk[k++] = t;
    *n_unk = o.unknown.size();
    *flags = (o.empty_input ? 1 : 0) | (o.may_panic ? 2 : 0);
    return 0;
}

// Batch taxa2agg record loop (src/commands/taxa2agg.rs:161-180) over CSR input, checking a
// candidate result (got_taxon/got_status, e.g. from the GPU) for membership in the set of
// results the reference could produce.  Returns the number of mismatching records; the first
// `cap_bad` mismatching record indices go to bad[].  If want_taxon != NULL it receives the
// smallest member of each record's outcome set (status in want_status) — for reporting.
size_t orc_taxa2agg_check(void* h, int ranked, float lower_bound, const uint32_t* taxa, const float* scores,
                          const uint64_t* offsets, size_t n_reads, const uint32_t* got_taxon,
                          const uint32_t* got_status, uint64_t* bad, size_t cap_bad, uint32_t* want_taxon,
                          uint32_t* want_status, int nthreads) {
    AggHandle* ah = (AggHandle*)h;
    std::vector<std::optional<TaxonId>> snapping = ah->tx->snapping(ranked != 0);
    std::atomic<size_t> nbad{0};
    std::vector<std::vector<uint64_t>> bad_local((size_t)std::max(1, nthreads));
    auto work = [&](size_t tid, size_t lo, size_t hi) {
        std::vector<std::pair<TaxonId, float>> items;
        for (size_t r = lo; r < hi; r++) {
            items.clear();
            for (uint64_t j = offsets[r]; j < offsets[r + 1]; j++)
                items.emplace_back((TaxonId)taxa[j], scores ? scores[j] : 1.0f);
            RecordOutcome ro;
            try { ro = taxa2agg_record(*ah->agg, snapping, items, lower_bound); }
            catch (Panic&) { ro.may_panic = true; }
            bool ok = false;
            uint32_t wt = 0, ws = ST_OK;
            if (ro.empty) { wt = 1; ws = ST_EMPTY; }
            else if (!ro.unknown.empty()) { wt = (uint32_t)*ro.unknown.begin(); ws = ST_UNKNOWN_TAXON; }
            else if (ro.may_panic && ro.printed.empty()) { wt = 0; ws = ST_PANIC; }
            else if (!ro.printed.empty()) { wt = (uint32_t)*ro.printed.begin(); ws = ST_OK; }
            if (got_taxon && got_status) {
                uint32_t gs = got_status[r], gt = got_taxon[r];
                if (ro.empty) ok = (gs == ST_EMPTY && gt == 1);
                else if (!ro.unknown.empty() || ro.may_panic) {
                    // an erroring record: any of the reachable errors is acceptable; if some orders
                    // print a value while others panic (snapping None on a tie member) accept both.
                    if (gs == ST_UNKNOWN_TAXON) ok = ro.unknown.count(gt) > 0;
                    else if (gs == ST_PANIC) ok = ro.may_panic;
                    else if (gs == ST_OK) ok = ro.unknown.empty() && ro.printed.count(gt) > 0;
                } else ok = (gs == ST_OK && ro.printed.count(gt) > 0);
                if (!ok) { nbad++; bad_local[tid].push_back(r); }
            }
            if (want_taxon) want_taxon[r] = wt;
            if (want_status) want_status[r] = ws;
        }
    };
    size_t T = (size_t)std::max(1, nthreads);
    std::vector<std::thread> th;
    for (size_t t = 0; t < T; t++) {
        size_t lo = n_reads * t / T, hi = n_reads * (t + 1) / T;
        th.emplace_back(work, t, lo, hi);
    }
    for (auto& t : th) t.join();
    size_t k = 0;
    for (auto& bl : bad_local) for (auto b : bl) if (k < cap_bad) bad[k++] = b;
    std::sort(bad, bad + k);
    return nbad.load();
}

// ---- seedextend on arrays (src/commands/seedextend.rs:99-148), for device parity ----------
// in: taxa of one record (without the appended sentinel); out: kept taxa; returns count, or (size_t)-1 where the
// reference panics (a seed sliced with start > end, src/commands/seedextend.rs:169-173)
size_t orc_seedextend_record(const uint32_t* taxa, size_t n, size_t min_seed_size, size_t max_gap_size,
                             uint32_t* out) {
    std::ostringstream in_s, out_s;
    in_s << ">r\n";
    for (size_t i = 0; i < n; i++) in_s << taxa[i] << "\n";
    std::istringstream is(in_s.str());
    try {
        seedextend(is, out_s, min_seed_size, max_gap_size, nullptr, 5);
    } catch (const Panic&) {
        return (size_t)-1;
    }
    std::istringstream os(out_s.str());
    std::string line;
    size_t k = 0;
    std::getline(os, line);  // header
    while (std::getline(os, line)) out[k++] = (uint32_t)parse_usize(line);
    return k;
}

// ranked mode (`-r`, src/commands/seedextend.rs:150-163) on one record; tax = orc_taxonomy_* handle.  overflow_checks != 0 models a
// build with overflow checks (the `e - s` of an inverted seed panics), 0 a release build (it wraps).
void* orc_ranked_list_new(void* tax) {
    return new TaxonList(TaxonList::new_with_unknown(((Taxonomy*)tax)->taxons, true));
}
void orc_ranked_list_free(void* h) { delete (TaxonList*)h; }
size_t orc_seedextend_record_ranked(void* ranked_list, const uint32_t* taxa, size_t n, size_t min_seed_size, size_t max_gap_size,
                                    size_t penalty, int overflow_checks, uint32_t* out) {
    std::ostringstream in_s, out_s;
    in_s << ">r\n";
    for (size_t i = 0; i < n; i++) in_s << taxa[i] << "\n";
    std::istringstream is(in_s.str());
    try {
        seedextend(is, out_s, min_seed_size, max_gap_size, (const TaxonList*)ranked_list, penalty, overflow_checks != 0);
    } catch (const Panic&) {
        return (size_t)-1;
    }
    std::istringstream os(out_s.str());
    std::string line;
    size_t k = 0;
    std::getline(os, line);  // header
    while (std::getline(os, line)) out[k++] = (uint32_t)parse_usize(line);
    return k;
}

// ---- synthetic closed-form index (tests + bench cpu_baseline) ------------------------------
// 9-mer lookup against the closed-form synthetic index; miss => miss_value
void orc_synth_lookup9(const SynthIndexParams* p, const uint32_t* taxon_of_row, const uint64_t* keys45, size_t n,
                       uint32_t miss_value, uint32_t* out) {
    uint64_t ainv = synth_modinv(p->mul_a % SYNTH_M, SYNTH_M);
    for (size_t i = 0; i < n; i++) {
        uint64_t member;
        if (synth_index_member(*p, ainv, keys45[i], &member)) out[i] = taxon_of_row[member % p->n_tax];
        else out[i] = miss_value;
    }
}
uint64_t orc_synth_key_of_member(const SynthIndexParams* p, uint64_t member) { return synth_key_of_member(*p, member); }

// ---- CPU baseline: sorted-array index, binary search (dependent accesses), threads -------
struct SortedU64Index { std::vector<uint64_t> keys; std::vector<uint32_t> vals; };
void* orc_sorted_index_new(const uint64_t* keys_sorted, const uint32_t* vals, size_t n) {
    auto* s = new SortedU64Index();
    s->keys.assign(keys_sorted, keys_sorted + n);
    s->vals.assign(vals, vals + n);
    return s;
}
void orc_sorted_index_free(void* h) { delete (SortedU64Index*)h; }
void orc_sorted_lookup9(void* h, const uint64_t* keys, size_t n, uint32_t miss_value, uint32_t* out, int nthreads) {
    SortedU64Index* s = (SortedU64Index*)h;
    size_t T = (size_t)std::max(1, nthreads);
    std::vector<std::thread> th;
    for (size_t t = 0; t < T; t++) {
        size_t lo = n * t / T, hi = n * (t + 1) / T;
        th.emplace_back([=]() {
            for (size_t i = lo; i < hi; i++) {
                auto it = std::lower_bound(s->keys.begin(), s->keys.end(), keys[i]);
                out[i] = (it != s->keys.end() && *it == keys[i]) ? s->vals[(size_t)(it - s->keys.begin())] : miss_value;
            }
        });
    }
    for (auto& t : th) t.join();
}

}  // extern "C"
