// main.cpp — the `umgap` command over libumgap_hip.so: same subcommand surface, flags, FASTA
// stdin/stdout bytes, messages and exit codes as the reference for the pept2lca -> taxa2agg path,
// so scripts/umgap-analyse.sh can call it unchanged.  Mirrors src/main.rs:8-63 (dispatch, exit
// codes) and the `Args` structs of src/commands/{pept2lca,prot2kmer2lca,taxa2agg,seedextend,uniq,
// buildindex}.rs.  All compute goes through the C ABI (include/umgap_hip.h): no GPU => exit 1.
//
// Output order: the reference emits rayon chunks in completion order (src/commands/pept2lca.rs:62-68);
// this command always emits input order (= the reference under RAYON_NUM_THREADS=1).
#include <sys/socket.h>
#include <sys/un.h>
#include <unistd.h>

#include <algorithm>
#include <cstdlib>
#include <functional>
#include <map>

#include "../../../include/umgap_hip.h"
#include "text.hpp"

using namespace umgap_host;

namespace {

[[noreturn]] void lib_fail(int rc) {
    (void)rc;
    throw CliError(umgap_last_error());
}
#define LIB(expr)                \
    do {                         \
        int rc__ = (expr);       \
        if (rc__) lib_fail(rc__); \
    } while (0)

// ---- clap-2 style argument parsing: -o, -om, -c N, -cN, --chunksize N, --chunksize=N -------
struct OptSpec {
    char shrt;
    const char* lng;
    bool takes_value;
};
struct Parsed {
    std::map<std::string, std::vector<std::string>> values;  // by long name
    std::vector<std::string> positional;
    bool has(const std::string& n) const { return values.count(n) > 0; }
    std::optional<std::string> get(const std::string& n) const {
        auto it = values.find(n);
        if (it == values.end() || it->second.empty()) return std::nullopt;
        return it->second.back();
    }
};
[[noreturn]] void usage_error(const std::string& msg) {
    fprintf(stderr, "error: %s\n\nFor more information try --help\n", msg.c_str());
    exit(1);
}
Parsed parse_args(const std::vector<std::string>& args, const std::vector<OptSpec>& specs) {
    Parsed p;
    auto by_long = [&](const std::string& n) -> const OptSpec* {
        for (auto& s : specs) if (n == s.lng) return &s;
        return nullptr;
    };
    auto by_short = [&](char c) -> const OptSpec* {
        for (auto& s : specs) if (c == s.shrt) return &s;
        return nullptr;
    };
    bool only_positional = false;
    for (size_t i = 0; i < args.size(); i++) {
        const std::string& a = args[i];
        if (only_positional || a.size() < 2 || a[0] != '-') { p.positional.push_back(a); continue; }
        if (a == "--") { only_positional = true; continue; }
        if (a[1] == '-') {
            std::string name = a.substr(2), val;
            bool has_val = false;
            size_t eq = name.find('=');
            if (eq != std::string::npos) { val = name.substr(eq + 1); name = name.substr(0, eq); has_val = true; }
            const OptSpec* s = by_long(name);
            if (!s) usage_error("Found argument '--" + name + "' which wasn't expected, or isn't valid in this context");
            if (s->takes_value) {
                if (!has_val) {
                    if (i + 1 >= args.size()) usage_error("The argument '--" + name + " <" + name + ">' requires a value but none was supplied");
                    val = args[++i];
                }
                p.values[s->lng].push_back(val);
            } else {
                if (has_val) usage_error("The argument '--" + name + "' takes no value");
                p.values[s->lng].push_back("");
            }
            continue;
        }
        for (size_t k = 1; k < a.size(); k++) {
            const OptSpec* s = by_short(a[k]);
            if (!s) usage_error(std::string("Found argument '-") + a[k] + "' which wasn't expected, or isn't valid in this context");
            if (s->takes_value) {
                std::string val;
                if (k + 1 < a.size()) { val = a.substr(k + 1); if (!val.empty() && val[0] == '=') val = val.substr(1); }
                else {
                    if (i + 1 >= args.size()) usage_error(std::string("The argument '-") + a[k] + "' requires a value but none was supplied");
                    val = args[++i];
                }
                p.values[s->lng].push_back(val);
                break;
            }
            p.values[s->lng].push_back("");
        }
    }
    return p;
}
std::string need_positional(const Parsed& p, const char* what) {
    if (p.positional.size() < 1) usage_error(std::string("The following required arguments were not provided:\n    <") + what + ">");
    if (p.positional.size() > 1) usage_error("Found argument '" + p.positional[1] + "' which wasn't expected, or isn't valid in this context");
    return p.positional[0];
}
uint64_t opt_usize(const Parsed& p, const char* name, uint64_t dflt) {
    auto v = p.get(name);
    if (!v) return dflt;
    try { return parse_usize(*v); } catch (CliError& e) { usage_error("Invalid value for '--" + std::string(name) + "': " + e.what()); }
}
float opt_f32(const Parsed& p, const char* name, float dflt) {
    auto v = p.get(name);
    if (!v) return dflt;
    try { return parse_f32(*v); } catch (CliError& e) { usage_error("Invalid value for '--" + std::string(name) + "': " + e.what()); }
}

struct Out {
    FILE* f;
    std::string buf;
    explicit Out(FILE* f_) : f(f_) { buf.reserve(1 << 22); }
    void flush() {
        if (!buf.empty()) { fwrite(buf.data(), 1, buf.size(), f); buf.clear(); }
        fflush(f);
    }
    void maybe_flush() { if (buf.size() > (1u << 22)) { fwrite(buf.data(), 1, buf.size(), f); buf.clear(); } }
};

void append_u64(std::string& s, uint64_t v) {
    char tmp[24];
    int n = 0;
    do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (n) s.push_back(tmp[--n]);
}

// Looks up a batch of byte strings in either kind of index.  9-mer table: strings that are not
// [A-Z]{9} miss without touching the device.  Result 0 = miss.
struct StringBatch {
    std::vector<uint8_t> bytes;
    std::vector<uint64_t> offsets{0};
    void add(std::string_view s) { bytes.insert(bytes.end(), s.begin(), s.end()); offsets.push_back(bytes.size()); }
    size_t size() const { return offsets.size() - 1; }
    void clear() { bytes.clear(); offsets.assign(1, 0); }
};
void lookup_strings(umgap_index* ix, const StringBatch& b, std::vector<uint32_t>& out) {
    size_t n = b.size();
    out.assign(n, 0u);
    if (!n) return;
    if (umgap_index_kind(ix) == 2) {
        LIB(umgap_index_lookup_var(ix, b.bytes.data(), b.offsets.data(), n, 0u, out.data()));
        return;
    }
    std::vector<uint64_t> keys;
    std::vector<size_t> where;
    for (size_t i = 0; i < n; i++) {
        uint64_t key;
        if (b.offsets[i + 1] - b.offsets[i] == 9 && umgap_pack9((const char*)b.bytes.data() + b.offsets[i], &key)) { keys.push_back(key); where.push_back(i); }
    }
    std::vector<uint32_t> t(keys.size());
    if (!keys.empty()) LIB(umgap_index_lookup9(ix, keys.data(), keys.size(), 0u, t.data()));
    for (size_t k = 0; k < keys.size(); k++) out[where[k]] = t[k];
}

// ---- pept2lca / prot2kmer2lca ---------------------------------------------------------------
// One batch = many records: pack every peptide, ONE device lookup, then format in input order.
void stream_lookup(FILE* in, FILE* outf, umgap_index* ix, bool one_on_one, bool kmer_mode, size_t k) {
    // peptides other than [A-Z]{9} cannot be in a 9-mer table: they miss without touching the device
    FastaReader reader(in, kmer_mode);
    Out out(outf);
    const size_t BATCH_KEYS = 8u << 20;
    std::vector<Record> recs;
    std::vector<uint64_t> keys;
    std::vector<uint32_t> taxa;
    std::vector<uint8_t> known;  // per item/window: 1 = looked up on the device
    auto run = [&]() {
        taxa.resize(keys.size());
        if (!keys.empty()) LIB(umgap_index_lookup9(ix, keys.data(), keys.size(), 0u, taxa.data()));
        size_t ki = 0, wi = 0;
        for (auto& r : recs) {
            if (kmer_mode) {
                // src/commands/prot2kmer2lca.rs:172: reads shorter than k vanish, header included
                const std::string& prot = r.sequence[0];
                if (prot.size() < k) continue;
                out.buf.push_back('>'); out.buf.append(r.header); out.buf.push_back('\n');
                for (size_t i = 0; i + k <= prot.size(); i++, wi++) {
                    uint32_t t = known[wi] ? taxa[ki++] : 0u;
                    if (t != 0 || one_on_one) { append_u64(out.buf, t); out.buf.push_back('\n'); }
                }
            } else {
                out.buf.push_back('>'); out.buf.append(r.header); out.buf.push_back('\n');  // src/commands/pept2lca.rs:91
                for (size_t i = 0; i < r.sequence.size(); i++, wi++) {
                    uint32_t t = known[wi] ? taxa[ki++] : 0u;
                    if (t != 0 || one_on_one) { append_u64(out.buf, t); out.buf.push_back('\n'); }  // :93-95
                }
            }
            out.maybe_flush();
        }
        recs.clear(); keys.clear(); known.clear();
    };
    Record rec;
    try {
        while (reader.read_record(rec)) {
            if (kmer_mode) {
                const std::string& prot = rec.sequence[0];
                if (prot.size() >= k) {
                    for (size_t i = 0; i + k <= prot.size(); i++) {
                        // `&prot[i..i+k]` panics when it cuts a UTF-8 character (src/commands/prot2kmer2lca.rs:176)
                        if (((unsigned char)prot[i] & 0xC0) == 0x80 || (i + k < prot.size() && ((unsigned char)prot[i + k] & 0xC0) == 0x80)) {
                            run(); out.flush();
                            throw CliPanic("byte index is not a char boundary");
                        }
                        uint64_t key;
                        bool ok = k == 9 && umgap_pack9(prot.data() + i, &key);
                        known.push_back(ok);
                        if (ok) keys.push_back(key);
                    }
                }
            } else {
                for (auto& s : rec.sequence) {
                    uint64_t key;
                    bool ok = s.size() == 9 && umgap_pack9(s.data(), &key);
                    known.push_back(ok);
                    if (ok) keys.push_back(key);
                }
            }
            recs.push_back(std::move(rec));
            rec = Record();
            if (keys.size() >= BATCH_KEYS || recs.size() >= (1u << 20)) run();
        }
    } catch (CliError&) {
        run(); out.flush();  // records before the failing one are still printed
        throw;
    }
    run();
    out.flush();
}

umgap_index* open_index(const std::string& path) {
    umgap_index* ix = nullptr;
    int rc = umgap_index_open(path.c_str(), 1u, &ix);
    if (rc) lib_fail(rc);
    return ix;
}

// record loop shared by pept2lca (variable-length index) and prot2tryp2lca: `expand` turns one
// sequence line into the peptides to look up
void stream_strings(FILE* in, FILE* outf, umgap_index* ix, bool one_on_one,
                    const std::function<void(const std::string&, StringBatch&)>& expand) {
    FastaReader reader(in, false);
    Out out(outf);
    std::vector<std::string> headers;
    std::vector<size_t> first;  // index of the record's first peptide in the batch
    StringBatch batch;
    std::vector<uint32_t> taxa;
    auto run = [&]() {
        lookup_strings(ix, batch, taxa);
        for (size_t r = 0; r < headers.size(); r++) {
            out.buf.push_back('>'); out.buf.append(headers[r]); out.buf.push_back('\n');
            size_t end = r + 1 < headers.size() ? first[r + 1] : batch.size();
            for (size_t i = first[r]; i < end; i++)
                if (taxa[i] != 0 || one_on_one) { append_u64(out.buf, taxa[i]); out.buf.push_back('\n'); }
            out.maybe_flush();
        }
        headers.clear(); first.clear(); batch.clear();
    };
    Record rec;
    try {
        while (reader.read_record(rec)) {
            first.push_back(batch.size());
            headers.push_back(std::move(rec.header));
            for (auto& s : rec.sequence) expand(s, batch);
            if (batch.bytes.size() >= (64u << 20) || headers.size() >= (1u << 20)) run();
        }
    } catch (CliError&) { run(); out.flush(); throw; }
    run();
    out.flush();
}

int cmd_pept2lca(const std::vector<std::string>& args) {
    Parsed p = parse_args(args, {{'o', "one-on-one", false}, {'m', "in-memory", false}, {'c', "chunksize", true}});
    (void)opt_usize(p, "chunksize", 240);  // chunks only group output in the reference; accepted and validated
    std::string path = need_positional(p, "fst-file");
    umgap_index* ix = open_index(path);  // -m: the table is always resident (in HBM)
    if (umgap_index_kind(ix) == 2) stream_strings(stdin, stdout, ix, p.has("one-on-one"), [](const std::string& s, StringBatch& b) { b.add(s); });
    else stream_lookup(stdin, stdout, ix, p.has("one-on-one"), false, 9);
    umgap_index_free(ix);
    return 0;
}

// prot2tryp2lca: src/commands/prot2tryp2lca.rs:88-140.  The default pattern "([KR])([^P])" applied twice
// and '*' -> newline is: cut after every K/R that is not followed by P; '*' separates and is dropped.
int cmd_prot2tryp2lca(const std::vector<std::string>& args) {
    Parsed p = parse_args(args, {{'o', "one-on-one", false}, {'m', "in-memory", false}, {'c', "chunksize", true}, {'p', "pattern", true},
                                 {'l', "minlen", true}, {'L', "maxlen", true}, {'k', "keep", true}, {'d', "drop", true}});
    (void)opt_usize(p, "chunksize", 240);
    size_t minlen = (size_t)opt_usize(p, "minlen", 5), maxlen = (size_t)opt_usize(p, "maxlen", 50);
    std::string keep = p.get("keep").value_or(""), drop = p.get("drop").value_or("");
    if (p.get("pattern").value_or("([KR])([^P])") != "([KR])([^P])")
        throw CliError("only the default cleavage pattern ([KR])([^P]) is supported by the accelerated prot2tryp2lca");
    std::string path = need_positional(p, "fst-file");
    umgap_index* ix = open_index(path);
    auto expand = [&](const std::string& seq, StringBatch& b) {
        size_t st = 0;
        auto emit = [&](size_t a, size_t e) {
            std::string_view pep(seq.data() + a, e - a);
            if (!pep.empty() && pep.back() == '\r') pep.remove_suffix(1);   // str::lines() strips a trailing CR
            if (pep.empty() || pep.size() < minlen || pep.size() > maxlen) return;
            if (!keep.empty() || !drop.empty()) {
                for (char c : keep) if (pep.find(c) == std::string_view::npos) return;
                for (char c : drop) if (pep.find(c) != std::string_view::npos) return;
            }
            b.add(pep);
        };
        for (size_t i = 0; i < seq.size(); i++) {
            char c = seq[i];
            if (c == '*' || c == '\n') { emit(st, i); st = i + 1; }
            else if ((c == 'K' || c == 'R') && i + 1 < seq.size() && seq[i + 1] != 'P') { emit(st, i + 1); st = i + 1; }
        }
        emit(st, seq.size());
    };
    stream_strings(stdin, stdout, ix, p.has("one-on-one"), expand);
    umgap_index_free(ix);
    return 0;
}

int cmd_prot2kmer2lca(const std::vector<std::string>& args) {
    Parsed p = parse_args(args, {{'k', "length", true}, {'o', "one-on-one", false}, {'s', "socket", true}, {'m', "in-memory", false}, {'c', "chunksize", true}});
    size_t k = (size_t)opt_usize(p, "length", 9);
    (void)opt_usize(p, "chunksize", 240);
    std::string path = need_positional(p, "fst-file");
    umgap_index* ix = open_index(path);
    if (umgap_index_kind(ix) != 9) throw CliError("prot2kmer2lca needs a 9-mer index");
    bool o = p.has("one-on-one");
    if (auto sock = p.get("socket")) {
        // src/commands/prot2kmer2lca.rs:116-137: sequential connections, status lines on stdout
        int fd = socket(AF_UNIX, SOCK_STREAM, 0);
        if (fd < 0) throw CliError(strerror(errno));
        sockaddr_un addr{};
        addr.sun_family = AF_UNIX;
        if (sock->size() >= sizeof(addr.sun_path)) throw CliError("path must be shorter than SUN_LEN");
        strcpy(addr.sun_path, sock->c_str());
        if (bind(fd, (sockaddr*)&addr, sizeof(addr)) != 0 || listen(fd, 128) != 0) throw CliError(std::string(strerror(errno)) + " (os error " + std::to_string(errno) + ")");
        printf("Socket created, listening for connections.\n"); fflush(stdout);
        for (;;) {
            int c = accept(fd, nullptr, nullptr);
            if (c < 0) continue;
            printf("Connection accepted. Processing...\n"); fflush(stdout);
            FILE* in = fdopen(c, "rb");
            FILE* outf = fdopen(dup(c), "wb");
            try {
                stream_lookup(in, outf, ix, o, true, k);
                printf("Connection finished succesfully.\n");
            } catch (CliError& e) { printf("Connection died with an error: %s\n", e.what()); }
            fflush(stdout);
            fclose(outf);
            fclose(in);
        }
    }
    stream_lookup(stdin, stdout, ix, o, true, k);
    umgap_index_free(ix);
    return 0;
}

// ---- taxa2agg ------------------------------------------------------------------------------------
umgap_taxonomy* load_taxonomy(const std::string& path) {
    TaxonRows rows = read_taxa_file(path);
    std::vector<uint32_t> id(rows.id.size()), parent(rows.id.size());
    for (size_t i = 0; i < rows.id.size(); i++) {
        if (rows.id[i] >= 0xFFFFFFFFull || rows.parent[i] >= 0xFFFFFFFFull) throw CliError("taxon id >= 2^32-1 is not supported");
        id[i] = (uint32_t)rows.id[i];
        parent[i] = (uint32_t)rows.parent[i];
    }
    umgap_taxonomy* tx = nullptr;
    int rc = umgap_taxonomy_create(id.data(), parent.data(), rows.valid.data(), rows.rank.data(), id.size(), 0, &tx);
    if (rc == UMGAP_ERR_PANIC) throw CliPanic(umgap_last_error());
    if (rc) lib_fail(rc);
    return tx;
}

int cmd_taxa2agg(const std::vector<std::string>& args) {
    Parsed p = parse_args(args, {{'s', "scored", false}, {'r', "ranked", false}, {'m', "method", true}, {'a', "aggregate", true},
                                 {'f', "factor", true}, {'l', "lower-bound", true}});
    std::string m = p.get("method").value_or("tree"), a = p.get("aggregate").value_or("hybrid");
    if (m != "tree" && m != "rmq") usage_error("'" + m + "' isn't a valid value for '--method <method>'\n\t[possible values: rmq, tree]");
    if (a != "lca*" && a != "hybrid" && a != "mrtl") usage_error("'" + a + "' isn't a valid value for '--aggregate <strategy>'\n\t[possible values: hybrid, lca*, mrtl]");
    float factor = opt_f32(p, "factor", 0.25f), lower_bound = opt_f32(p, "lower-bound", 0.0f);
    bool scored = p.has("scored"), ranked = p.has("ranked");
    std::string path = need_positional(p, "taxon-file");
    int method = m == "tree" ? UMGAP_METHOD_TREE : UMGAP_METHOD_RMQ;
    int strategy = a == "lca*" ? UMGAP_STRATEGY_LCA_STAR : (a == "hybrid" ? UMGAP_STRATEGY_HYBRID : UMGAP_STRATEGY_MRTL);
    umgap_taxonomy* tx = load_taxonomy(path);  // parse errors come first, like src/commands/taxa2agg.rs:104
    if (method == UMGAP_METHOD_RMQ && strategy == UMGAP_STRATEGY_HYBRID)
        fprintf(stderr, "Warning: this is a hybrid between LCA/MRTL, not LCA*/MRTL\n");  // :119-123
    if (method == UMGAP_METHOD_TREE && strategy == UMGAP_STRATEGY_MRTL)
        throw CliError("Invalid invocation: Tree and MaximumRootToLeafPath cannot be combined");  // :134-138

    FastaReader reader(stdin, false);
    Out out(stdout);
    std::vector<std::string> headers;
    std::vector<uint32_t> taxa, out_taxon, out_status;
    std::vector<uint64_t> big;  // original values of ids that do not fit 32 bits (for the message)
    std::vector<float> scores;
    std::vector<uint64_t> offsets{0};
    std::optional<CliError> pending_error;
    auto run = [&]() {
        size_t n = headers.size();
        out_taxon.resize(n); out_status.resize(n);
        if (n) LIB(umgap_aggregate(tx, method, strategy, factor, lower_bound, ranked ? 1 : 0, taxa.data(), scored ? scores.data() : nullptr,
                                   offsets.data(), n, out_taxon.data(), out_status.data()));
        for (size_t r = 0; r < n; r++) {
            if (out_status[r] == UMGAP_READ_UNKNOWN_TAXON) {
                out.flush();
                uint64_t t = out_taxon[r];
                if (t == 0xFFFFFFFEu && !big.empty()) t = big[0];
                throw CliError("Unknown Taxon ID: " + std::to_string(t));
            }
            if (out_status[r] == UMGAP_READ_PANIC) { out.flush(); throw CliPanic("index out of bounds"); }
            out.buf.push_back('>'); out.buf.append(headers[r]); out.buf.push_back('\n');
            append_u64(out.buf, out_taxon[r]);
            out.buf.push_back('\n');
            out.maybe_flush();
        }
        headers.clear(); taxa.clear(); scores.clear(); big.clear(); offsets.assign(1, 0);
    };
    Record rec;
    try {
        while (reader.read_record(rec)) {
            for (auto& s : rec.sequence) {
                uint64_t t;
                float sc = 1.0f;
                if (scored) {  // with_score, src/commands/taxa2agg.rs:142-148
                    size_t eq = s.find('=');
                    if (eq == std::string::npos || s.find('=', eq + 1) != std::string::npos) throw CliError("Taxon without score");
                    t = parse_usize(std::string_view(s).substr(0, eq));
                    sc = parse_f32(std::string_view(s).substr(eq + 1));
                    scores.push_back(sc);
                } else {
                    t = parse_usize(s);
                }
                if (t >= 0xFFFFFFFEull) { big.push_back(t); t = 0xFFFFFFFEull; }
                taxa.push_back((uint32_t)t);
            }
            offsets.push_back(taxa.size());
            headers.push_back(std::move(rec.header));
            if (taxa.size() >= (16u << 20) || headers.size() >= (1u << 20)) run();
        }
    } catch (CliError& e) {
        pending_error = e;
        // the failing record is dropped; everything before it is still aggregated and printed
        taxa.resize(offsets.back());
        if (scored) scores.resize(offsets.back());
    }
    run();
    out.flush();
    umgap_taxonomy_free(tx);
    if (pending_error) throw *pending_error;
    return 0;
}

// ---- seedextend (unranked mode on the device) -------------------------------------------------
int cmd_seedextend(const std::vector<std::string>& args) {
    Parsed p = parse_args(args, {{'s', "min-seed-size", true}, {'g', "max-gap-size", true}, {'r', "ranked", true}, {'p', "penalty", true}});
    uint64_t s = opt_usize(p, "min-seed-size", 2), g = opt_usize(p, "max-gap-size", 0);
    (void)opt_usize(p, "penalty", 5);
    if (p.has("ranked")) throw CliError("seedextend --ranked is outside the accelerated path (scripts/umgap-analyse.sh never uses it)");
    if (!p.positional.empty()) usage_error("Found argument '" + p.positional[0] + "' which wasn't expected, or isn't valid in this context");
    // host restatement of the record state machine for the text filter; the device kernel
    // (umgap_seedextend_dev) serves the fused pipeline.  src/commands/seedextend.rs:99-148
    FastaReader reader(stdin, false);
    Out out(stdout);
    Record rec;
    std::vector<uint64_t> t;
    std::vector<std::string> items;
    try {
        while (reader.read_record(rec)) {
            t.clear();
            for (auto& x : rec.sequence) t.push_back(parse_usize(x));
            t.push_back(0);
            std::vector<std::pair<size_t, size_t>> seeds;
            size_t start = 0, end = 1, same_tid = 1, same_max = 1;
            uint64_t last = t[0];
            while (end < t.size()) {
                if (last == t[end]) { same_tid++; end++; continue; }
                if (last == 0 && same_tid > g) {
                    if (same_max >= s) seeds.emplace_back(start, end - same_tid);
                    start = end; last = t[end]; same_tid = 1; same_max = 1; end++;
                    continue;
                }
                if (last == 0 && (end - start) == same_tid) { end++; start = end; continue; }
                if (last != 0) same_max = std::max(same_max, same_tid);
                last = t[end]; same_tid = 1; end++;
            }
            if (same_max >= s) {
                if (last == 0) end -= same_tid;
                seeds.emplace_back(start, end);
            }
            items.clear();
            for (auto& se : seeds) {
                if (se.first > se.second || se.second > t.size()) { out.flush(); throw CliPanic("slice index out of range"); }
                for (size_t j = se.first; j < se.second; j++) items.push_back(std::to_string(t[j]));
            }
            write_record(out.buf, rec.header, items, "\n", false);
            out.maybe_flush();
        }
    } catch (CliError&) { out.flush(); throw; }
    out.flush();
    return 0;
}

// ---- uniq (src/commands/uniq.rs:56-84) ----------------------------------------------------------
int cmd_uniq(const std::vector<std::string>& args) {
    Parsed p = parse_args(args, {{'s', "separator", true}, {'w', "wrap", false}, {'d', "delimiter", true}});
    std::string sep = p.get("separator").value_or("\n");
    bool wrap = p.has("wrap");
    auto delim = p.get("delimiter");
    if (!p.positional.empty()) usage_error("Found argument '" + p.positional[0] + "' which wasn't expected, or isn't valid in this context");
    FastaReader reader(stdin, false);
    Out out(stdout);
    std::optional<Record> last;
    Record rec;
    try {
        while (reader.read_record(rec)) {
            if (delim) {
                size_t at = rec.header.find(*delim);
                if (at != std::string::npos) rec.header.resize(at);
            }
            if (last && last->header == rec.header) {
                for (auto& x : rec.sequence) last->sequence.push_back(std::move(x));
            } else {
                if (last) { write_record(out.buf, last->header, last->sequence, sep, wrap); out.maybe_flush(); }
                last = std::move(rec);
                rec = Record();
            }
        }
    } catch (CliError&) { out.flush(); throw; }
    if (last) write_record(out.buf, last->header, last->sequence, sep, wrap);
    out.flush();
    return 0;
}

// ---- buildindex: TSV (9-mer<TAB>taxon, sorted) on stdin -> UMGAPIX1 table on stdout ---------
int cmd_buildindex(const std::vector<std::string>& args) {
    Parsed p = parse_args(args, {{'l', "load-factor", true}});
    float load = opt_f32(p, "load-factor", 0.0f);
    LineSource ls(stdin);
    std::string_view l;
    std::vector<uint64_t> keys;
    std::vector<uint32_t> vals;
    StringBatch strs;
    bool all9 = true;
    std::string prev;
    while (ls.next(l)) {
        if (l.empty()) continue;
        size_t tab = l.find('\t');
        if (tab == std::string_view::npos || l.find('\t', tab + 1) != std::string_view::npos) throw CliError("CSV deserialize error: expected two fields");
        std::string_view k = l.substr(0, tab);
        uint64_t v;
        try { v = parse_usize(l.substr(tab + 1)); } catch (CliError&) { throw CliError("CSV deserialize error: invalid integer"); }
        // fst::MapBuilder::insert wants strictly increasing keys (src/commands/buildindex.rs:42)
        if (!prev.empty() && !(std::string_view(prev) < k)) throw CliError(std::string(prev == k ? "Error inserting duplicate key: " : "Error inserting out-of-order key: ") + std::string(k));
        prev.assign(k);
        uint64_t key = 0;
        if (k.size() != 9 || !umgap_pack9(k.data(), &key)) all9 = false;
        if (k.empty() || k.size() > 63) throw CliError("key '" + std::string(k) + "': keys must be 1..63 bytes long");
        if (v > 0xFFFFFFFFull) throw CliError("taxon id does not fit 32 bits");
        keys.push_back(key);
        strs.add(k);
        vals.push_back((uint32_t)v);
    }
    // every key [A-Z]{9} => the packed 9-mer table; anything else => the variable-length table
    umgap_index* ix = nullptr;
    if (all9) LIB(umgap_index_build9(keys.data(), vals.data(), keys.size(), load, 0, &ix));
    else LIB(umgap_index_build_var(strs.bytes.data(), strs.offsets.data(), vals.data(), vals.size(), load, 0, &ix));
    LIB(umgap_index_save(ix, "/dev/stdout"));
    umgap_index_free(ix);
    return 0;
}

}  // namespace

int main(int argc, char** argv) {
    std::vector<std::string> args(argv + 1, argv + argc);
    try {
        if (args.empty()) usage_error("umgap: a subcommand is required (pept2lca, prot2kmer2lca, taxa2agg, seedextend, uniq, buildindex)");
        std::string cmd = args[0];
        args.erase(args.begin());
        if (cmd == "-V" || cmd == "--version") { printf("umgap %s\n", umgap_version()); return 0; }  // scripts/umgap-analyse.sh:107
        if (cmd == "pept2lca") return cmd_pept2lca(args);
        if (cmd == "prot2kmer2lca") return cmd_prot2kmer2lca(args);
        if (cmd == "prot2tryp2lca") return cmd_prot2tryp2lca(args);
        if (cmd == "taxa2agg") return cmd_taxa2agg(args);
        if (cmd == "seedextend") return cmd_seedextend(args);
        if (cmd == "uniq") return cmd_uniq(args);
        if (cmd == "buildindex") return cmd_buildindex(args);
        usage_error("Found argument '" + cmd + "' which wasn't expected, or isn't valid in this context");
    } catch (CliError& e) {
        fflush(stdout);
        fprintf(stderr, "Error: %s\n", e.what());
        for (auto& c : e.causes) fprintf(stderr, "Caused by: %s\n", c.c_str());
        return 1;
    } catch (CliPanic& e) {
        fflush(stdout);
        fprintf(stderr, "thread 'main' panicked at '%s'\n", e.what());
        return 101;
    }
}
