// main.cpp — the `umgap` command over libumgap_hip.so: same subcommand surface, flags, FASTA
// stdin/stdout bytes, messages and exit codes as the reference for the pept2lca -> taxa2agg path,
// so scripts/umgap-analyse.sh can call it unchanged.  Mirrors src/main.rs:8-63 (dispatch, exit
// codes) and the `Args` structs of src/commands/{pept2lca,prot2kmer2lca,taxa2agg,seedextend,uniq,
// buildindex}.rs.  All compute goes through the C ABI (include/umgap_hip.h): no GPU => exit 1.
//
// Output order: the reference emits rayon chunks in completion order (src/commands/pept2lca.rs:62-68);
// this command always emits input order (= the reference under RAYON_NUM_THREADS=1).
#include <fcntl.h>
#include <signal.h>
#include <spawn.h>
#include <sys/mman.h>
#include <sys/prctl.h>
#include <sys/socket.h>
#include <sys/stat.h>
#include <sys/un.h>
#include <sys/wait.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <functional>
#include <map>
#include <atomic>
#include <memory>
#include <regex>

#include "../../../include/umgap_hip.h"
#include "async_io.hpp"
#include "cleave_pattern.hpp"
#include "fastpath.hpp"
#include "fuse_parse.hpp"
#include "text.hpp"

extern char** environ;

using namespace umgap_host;

namespace {
// an uninitialised, grow-only byte buffer
struct RawBuf {
    std::unique_ptr<char[]> p;
    size_t cap = 0, n = 0;
    void need(size_t bytes) { if (cap < bytes) { cap = bytes + bytes / 4; p.reset(new char[cap]); } n = 0; }
};

// Output that leaves without a copy: the bytes of a buffer are handed to the writer thread as they lie (AsyncIo::write_owned) and the
// latch says when the buffer is the command's own again.  Commands keep two sets of output buffers and alternate.
struct OutLatch {
    std::mutex m;
    std::condition_variable cv;
    size_t pending = 0;
    void add() { std::lock_guard<std::mutex> g(m); pending++; }
    void done() { std::lock_guard<std::mutex> g(m); if (--pending == 0) cv.notify_all(); }
    void wait() { std::unique_lock<std::mutex> lk(m); cv.wait(lk, [&] { return pending == 0; }); }
    ~OutLatch() { wait(); }            // (a buffer must outlive its write: declare the latch BEHIND the buffers it guards)
};
inline void write_owned_or_copy(FILE* f, const char* p, size_t n, OutLatch& latch) {
    if (!n) return;
    latch.add();
    if (!AsyncIo::write_owned(f, p, n, [&latch]() { latch.done(); })) {
        fwrite(p, 1, n, f);
        latch.done();
    }
}

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

void append_u64(std::string& s, uint64_t v) { append_uint(s, v); }

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

// Grow-only array in page-locked host memory (umgap_host_alloc): what umgap_classify_records copies from / to at full
// PCIe rate and without blocking the calling thread.  resize() keeps no old contents.
template <class T>
struct PinnedVec {
    T* p = nullptr;
    size_t n = 0, cap = 0;
    PinnedVec() = default;
    PinnedVec(const PinnedVec&) = delete;
    PinnedVec& operator=(const PinnedVec&) = delete;
    ~PinnedVec() { umgap_host_free(p); }
    void resize(size_t m) {
        if (m > cap) {
            umgap_host_free(p);
            p = nullptr;
            cap = 0;
            void* q = nullptr;
            const size_t want = m + m / 4 + 1024;
            if (umgap_host_alloc(want * sizeof(T), &q) != UMGAP_OK) throw CliError(umgap_last_error());
            p = (T*)q;
            cap = want;
        }
        n = m;
    }
    T* data() { return p; }
    T* begin() { return p; }
    size_t size() const { return n; }
    T& operator[](size_t i) { return p[i]; }
};


// ---- pept2lca / prot2kmer2lca on the packed 9-mer table -------------------------------------
// Block-parallel (fastpath.hpp): workers tokenize record-aligned ranges of a block and pack the
// 9-mers, ONE device lookup serves the block, workers format their ranges, output in input order.
// Peptides / windows that are not [A-Z]{9} cannot be in the table: they miss without a device call.
// (the per-range structs of the block-parallel commands are aligned to two cache lines: they sit side by side in one array, each is
// filled by its own thread with a push_back per item, and the vectors' end pointers live in the struct -- unaligned, neighbouring
// threads fought over those lines and 16 threads tokenized at a quarter of the single-thread rate each)
struct alignas(128) LookupPart {
    std::vector<size_t> hdr_off;
    std::vector<uint32_t> hdr_len, n_items;  // per record; n_items = items (pept2lca) or windows (prot2kmer2lca)
    std::vector<uint8_t> emit;               // per record: printed at all? (prot2kmer2lca drops reads shorter than k)
    std::vector<uint64_t> keys;              // packed keys of the `known` items, in order
    StringBatch strs;                        // string mode (a byte-string index): the items themselves, every one of them `known`
    std::vector<uint8_t> known;              // per item / window
    size_t key_base = 0;
    bool bad_utf8 = false, panic = false;    // the range stops at the record where this happened
    std::string out;
};

// string mode: what one sequence line of a record contributes to the batch (pept2lca on a byte-string index: the line itself;
// prot2tryp2lca: its tryptic peptides).  Called from several threads at once.
using LineExpand = std::function<void(std::string_view, StringBatch&)>;

void parse_lookup_part(const char* d, size_t begin, size_t end, bool kmer_mode, size_t k, LookupPart& P, const LineExpand* expand = nullptr) {
    LineCursor lc{d + begin, d + end};
    std::string_view line;
    std::string scratch;
    bool have_rec = false;
    // prot2kmer2lca: the record's sequence is the concatenation of its lines (unwrap, src/io/fasta.rs:62-64)
    std::string_view seq1;
    size_t n_lines = 0;
    auto finish_kmer_record = [&]() {
        if (!have_rec) return;
        std::string_view seq = n_lines <= 1 ? seq1 : std::string_view(scratch);
        if (seq.size() < k) { P.emit.back() = 0; return; }   // src/commands/prot2kmer2lca.rs:172
        uint32_t windows = (uint32_t)(seq.size() - k + 1);
        P.n_items.back() = windows;
        bool ascii = true;
        for (unsigned char c : seq) if (c >= 0x80) { ascii = false; break; }
        if (!ascii) {
            for (size_t i = 0; i + k <= seq.size(); i++)
                if (((unsigned char)seq[i] & 0xC0) == 0x80 || (i + k < seq.size() && ((unsigned char)seq[i + k] & 0xC0) == 0x80)) { P.panic = true; P.emit.back() = 0; P.n_items.back() = 0; return; }
        }
        if (expand) {      // a byte-string index serves every k (src/commands/prot2kmer2lca.rs:71-73): the windows themselves are looked up
            for (size_t i = 0; i + k <= seq.size(); i++) P.strs.add(seq.substr(i, k));
            P.known.insert(P.known.end(), windows, (uint8_t)1);
            return;
        }
        uint64_t key = 0;
        uint32_t run = 0;
        for (size_t i = 0; i < seq.size(); i++) {
            unsigned c = (unsigned char)seq[i] - 'A';
            if (c < 26) { key = ((key << 5) | c) & ((1ull << 45) - 1); run++; } else run = 0;
            if (i + 1 >= k) {
                bool ok = k == 9 && run >= 9;
                P.known.push_back(ok);
                if (ok) P.keys.push_back(key);
            }
        }
    };
    while (lc.next(line)) {
        bool utf8 = line_utf8_ok(line);
        if (!line.empty() && line[0] == '>' && utf8) {
            if (kmer_mode) { finish_kmer_record(); if (P.panic) return; }
            P.hdr_off.push_back((size_t)(line.data() - d) + 1);
            P.hdr_len.push_back((uint32_t)line.size() - 1);
            P.n_items.push_back(0);
            P.emit.push_back(1);
            have_rec = true;
            n_lines = 0;
            seq1 = std::string_view();   // a record without sequence lines has the empty sequence, not its predecessor's
            scratch.clear();
            continue;
        }
        if (!utf8) {  // the record read so far is emitted, then the stream errors out
            P.bad_utf8 = true;
            break;
        }
        if (kmer_mode) {
            if (n_lines == 0) seq1 = line;
            else { if (n_lines == 1) scratch.assign(seq1); scratch.append(line); }
            n_lines++;
        } else if (expand) {
            const size_t before = P.strs.size();
            (*expand)(line, P.strs);
            const size_t n = P.strs.size() - before;
            P.n_items.back() += (uint32_t)n;
            P.known.insert(P.known.end(), n, (uint8_t)1);
        } else {
            uint64_t key;
            bool ok = line.size() == 9 && umgap_pack9(line.data(), &key);
            P.known.push_back(ok);
            if (ok) P.keys.push_back(key);
            P.n_items.back()++;
        }
    }
    if (kmer_mode) finish_kmer_record();
}

// formats records [rb, re) of the part into `out`
void format_lookup_part(const char* d, const uint32_t* taxa, bool one_on_one, const LookupPart& P, size_t rb, size_t re, std::string& out) {
    size_t ki = 0, wi = 0;
    for (size_t r = 0; r < rb; r++)
        for (uint32_t i = 0; i < P.n_items[r] && wi < P.known.size(); i++, wi++) ki += P.known[wi];
    // consecutive windows of a protein mostly carry the same taxon: the line of the previous item is kept and appended again
    char last_line[24];
    size_t last_len = 0;
    uint32_t last_t = 0;
    for (size_t r = rb; r < re; r++) {
        if (P.emit[r]) { out.push_back('>'); out.append(d + P.hdr_off[r], P.hdr_len[r]); out.push_back('\n'); }
        for (uint32_t i = 0; i < P.n_items[r] && wi < P.known.size(); i++, wi++) {
            uint32_t t = P.known[wi] ? taxa[P.key_base + ki++] : 0u;
            if (P.emit[r] && (t != 0 || one_on_one)) {                                              // src/commands/pept2lca.rs:93-95
                if (t != last_t || !last_len) {
                    const size_t before = out.size();
                    append_uint(out, t);
                    out.push_back('\n');
                    last_len = out.size() - before;
                    memcpy(last_line, out.data() + before, last_len);
                    last_t = t;
                } else out.append(last_line, last_len);
            }
        }
    }
}

// Output follows the reference's chunking (src/io/fasta.rs:118-137, src/commands/pept2lca.rs:83-101): records
// are printed per chunk of `chunk` records, and a read error or a panic prints nothing of the chunk it falls
// in.  The partial chunk at the end of a block is therefore held back until the next block confirms it.
// get() over a batch of packed 9-mers: one GPU's table, or the table sharded over several
using LookupFn = std::function<void(const uint64_t* keys, size_t n, uint32_t* taxa)>;

// Two stages per block, overlapped: while block n is looked up on the device, formatted and written (stage B, on a helper
// thread), block n + 1 is read and tokenized (stage A).  Stage B runs for one block at a time and in input order, so the
// chunk bookkeeping (`held`, `g0`) and the order of output and errors are those of the plain loop.
// String mode (`expand` and `var_ix` given): the items are byte strings looked up in a byte-string index with ONE
// umgap_index_lookup_var per block; everything else -- ranges, chunks, errors, the two overlapped stages -- is the same.
void stream_lookup(FILE* in, FILE* outf, const LookupFn& lookup, bool one_on_one, bool kmer_mode, size_t k, size_t chunk, const LineExpand* expand = nullptr,
                   umgap_index* var_ix = nullptr) {
    if (chunk == 0) chunk = 1;
    BlockReader blocks(in);
    blocks.keep_previous_block();
    const unsigned T = worker_count();
    const char* d;
    size_t len;
    std::vector<uint64_t> keys;
    std::vector<uint8_t> str_bytes;        // string mode: the block's batch
    std::vector<uint64_t> str_offsets;
    std::vector<uint32_t> taxa;
    std::string held;          // formatted records [held_from, g0): the still open chunk
    size_t g0 = 0;             // global index of the block's first record
    struct Job { const char* d = nullptr; std::vector<LookupPart> parts; };
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const bool timing = getenv("UMGAP_TIMING") != nullptr;   // phase times on stderr
    double t_read = 0, t_parse = 0, t_wait = 0, t_keys = 0, t_lookup = 0, t_format = 0, t_write = 0;
    auto stage_b = [&](Job& job) {
        const char* d = job.d;
        std::vector<LookupPart>& parts = job.parts;
        const size_t np = parts.size();
        // a failing range ends the stream: later ranges are dropped
        size_t used = np;
        for (size_t i = 0; i < np; i++)
            if (parts[i].bad_utf8 || parts[i].panic) { used = i + 1; break; }
        const bool failed = parts[used - 1].bad_utf8 || parts[used - 1].panic;
        size_t total = 0, n_rec = 0;
        std::vector<size_t> rec_base(used + 1, 0);
        std::vector<uint64_t> byte_base(used + 1, 0);
        for (size_t i = 0; i < used; i++) {
            parts[i].key_base = total;
            total += expand ? parts[i].strs.size() : parts[i].keys.size();
            byte_base[i + 1] = byte_base[i] + parts[i].strs.bytes.size();
            rec_base[i] = n_rec;
            n_rec += parts[i].hdr_off.size();
        }
        rec_base[used] = n_rec;
        // records of this block to print now: global indices [g0, limit)
        size_t limit;
        if (failed) {
            // bad UTF-8: the error surfaces when the NEXT record is read, after n_rec good ones; panic: inside the last record
            size_t e = parts[used - 1].panic ? g0 + n_rec - 1 : g0 + n_rec;
            limit = e / chunk * chunk;
        } else {
            limit = (g0 + n_rec) / chunk * chunk;   // the open chunk at the end is held back (released at EOF below)
        }
        double t0 = now();
        if (expand) {       // the block's batch: the parts' strings back to back
            str_bytes.resize(byte_base[used] + 64);
            str_offsets.resize(total + 1);
            parallel_for(used, [&](size_t i) {
                const StringBatch& b = parts[i].strs;
                if (!b.bytes.empty()) memcpy(str_bytes.data() + byte_base[i], b.bytes.data(), b.bytes.size());
                for (size_t j = 0; j < b.size(); j++) str_offsets[parts[i].key_base + j] = byte_base[i] + b.offsets[j];
            });
            str_offsets[total] = byte_base[used];
        } else {
            keys.resize(total);
            parallel_for(used, [&](size_t i) { if (!parts[i].keys.empty()) memcpy(keys.data() + parts[i].key_base, parts[i].keys.data(), parts[i].keys.size() * 8); });
        }
        taxa.resize(total);
        t_keys += now() - t0; t0 = now();
        if (total) {
            if (expand) LIB(umgap_index_lookup_var(var_ix, str_bytes.data(), str_offsets.data(), total, 0u, taxa.data()));
            else lookup(keys.data(), total, taxa.data());
        }
        t_lookup += now() - t0; t0 = now();
        // per part: records below `limit` go to P.out, the rest to the tail (held back)
        std::vector<std::string> tails(used);
        parallel_for(used, [&](size_t i) {
            size_t nr = parts[i].hdr_off.size(), lo = g0 + rec_base[i];
            size_t cut = limit <= lo ? 0 : std::min(nr, limit - lo);
            parts[i].out.reserve(nr * 24 + parts[i].known.size() * 8);
            format_lookup_part(d, taxa.data(), one_on_one, parts[i], 0, cut, parts[i].out);
            if (cut < nr && !failed) format_lookup_part(d, taxa.data(), one_on_one, parts[i], cut, nr, tails[i]);
        });
        t_format += now() - t0; t0 = now();
        const size_t held_from = g0 / chunk * chunk;
        if (!held.empty() && limit > held_from) { fwrite(held.data(), 1, held.size(), outf); held.clear(); }
        for (size_t i = 0; i < used; i++) fwrite(parts[i].out.data(), 1, parts[i].out.size(), outf);
        t_write += now() - t0;
        if (failed) {
            fflush(outf);
            if (parts[used - 1].panic) throw CliPanic("byte index is not a char boundary");
            throw CliError("stream did not contain valid UTF-8");
        }
        for (size_t i = 0; i < used; i++) held.append(tails[i]);
        g0 += n_rec;
    };
    Job jobs[2];
    int cur = 0;
    std::thread helper;
    std::exception_ptr helper_err;
    auto join_helper = [&]() {
        if (helper.joinable()) helper.join();
        if (helper_err) { std::exception_ptr e = helper_err; helper_err = nullptr; std::rethrow_exception(e); }
    };
    try {
        for (;;) {
            double t0 = now();
            if (!blocks.next(d, len)) break;
            t_read += now() - t0; t0 = now();
            if (blocks.first_line_error()) { join_helper(); throw CliError("Expected > at beginning of fasta header."); }
            Job& J = jobs[cur];
            std::vector<size_t> cuts = split_records(d, len, T);
            const size_t np = cuts.size() - 1;
            J.d = d;
            J.parts.clear();
            J.parts.resize(np);
            parallel_for(np, [&](size_t i) { parse_lookup_part(d, cuts[i], cuts[i + 1], kmer_mode, k, J.parts[i], expand); });
            t_parse += now() - t0; t0 = now();
            join_helper();                         // stage B of the previous block is done (and has thrown here if it failed)
            t_wait += now() - t0;
            bool failing = false;
            for (const LookupPart& P : J.parts) failing = failing || P.bad_utf8 || P.panic;
            if (failing) { stage_b(J); continue; }  // the stream ends in this block: print what precedes the error and raise it now,
                                                    // without waiting for input that will never be used
            helper = std::thread([&stage_b, &helper_err, &J]() { try { stage_b(J); } catch (...) { helper_err = std::current_exception(); } });
            cur ^= 1;
        }
        join_helper();
    } catch (...) {
        if (helper.joinable()) helper.join();      // never leave the thread behind; its own error, if any, came second
        throw;
    }
    fwrite(held.data(), 1, held.size(), outf);   // EOF closes the last chunk
    fflush(outf);
    if (timing) fprintf(stderr, "stream_lookup: stage A: read %.3f, parse %.3f, wait for stage B %.3f | stage B: gather keys %.3f, lookup %.3f, format %.3f, write %.3f\n",
                        t_read, t_parse, t_wait, t_keys, t_lookup, t_format, t_write);
}

void stream_lookup(FILE* in, FILE* outf, umgap_index* ix, bool one_on_one, bool kmer_mode, size_t k, size_t chunk) {
    stream_lookup(in, outf, [ix](const uint64_t* keys, size_t n, uint32_t* taxa) { LIB(umgap_index_lookup9(ix, keys, n, 0u, taxa)); }, one_on_one, kmer_mode, k, chunk);
}

// `prot2kmer2lca` with k = 9 on a 9-mer index held by one GPU: the records' residues go to the device as they are and the windows
// are cut there (umgap_index_lookup_records) -- the host packs no keys and touches no window until it prints its hit.  Ranges,
// chunk bookkeeping (src/io/fasta.rs:118-137: output per completed chunk of `chunk` records, an error or a panic drops the open
// chunk), error order and the two overlapped stages are those of stream_lookup above, which keeps the other cases (any other k,
// a byte-string index, the table sharded over several GPUs).
void stream_kmer_records(FILE* in, FILE* outf, umgap_index* ix, bool one_on_one, size_t chunk) {
    if (chunk == 0) chunk = 1;
    BlockReader blocks(in);
    blocks.keep_previous_block();
    blocks.map_regular_file();
    if (in != stdin) blocks.connection();          // a connection of the -s server: blocks cut ahead by a filler thread, like stdin's
    const unsigned T = worker_count();
    struct Job {
        const char* d = nullptr;
        std::vector<KmerPart> parts;
        size_t np = 0;
        PinnedVec<uint8_t> bytes;
        PinnedVec<uint64_t> frag_off;
        PinnedVec<uint32_t> hits;
        std::vector<std::string> tails;
        OutLatch latch;        // the parts' output strings leave without a copy: they are this job's own again when the latch is down
                               // (declared behind `parts`: destroyed -- waited for -- before them)
    };
    std::string held;          // formatted records of the still open chunk
    size_t g0 = 0;             // global index of the block's first record
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const bool timing = getenv("UMGAP_TIMING") != nullptr;
    double t_read = 0, t_parse = 0, t_wait = 0, t_gather = 0, t_lookup = 0, t_format = 0, t_write = 0;
    // records [rb, re) of a part: header, then one line per window that hit (or every window with -o, misses as 0: src/commands/
    // prot2kmer2lca.rs:176-178); consecutive windows mostly carry the same taxon, whose line is kept and appended again
    auto format_part = [&](const char* d, const KmerPart& P, const uint32_t* hits, const uint64_t* frag_off, size_t rb, size_t re, std::string& out) {
        size_t f = P.rec_base;                     // fragment number (emitted records only) of record rb
        for (size_t r = 0; r < rb; r++) f += P.emit[r];
        // written through a raw pointer into a scratch buffer of the thread that is big enough for the range (no capacity check per
        // line), then appended to `out` in one piece
        size_t bound = 32;
        for (size_t r = rb; r < re; r++)
            if (P.emit[r]) bound += 2 + P.hdr_len[r] + ((size_t)P.seq_len[r] - 8) * 11;
        static thread_local RawBuf scratch;
        scratch.need(bound);
        char* w = scratch.p.get();
        char last_line[24] = {0};
        size_t last_len = 0;
        uint32_t last_t = 0;
        for (size_t r = rb; r < re; r++) {
            if (!P.emit[r]) continue;
            *w++ = '>';
            memcpy(w, d + P.hdr_off[r], P.hdr_len[r]);
            w += P.hdr_len[r];
            *w++ = '\n';
            const uint32_t* h = hits + frag_off[f];
            const size_t windows = (size_t)P.seq_len[r] - 8;
            f++;
            for (size_t i = 0; i < windows; i++) {
                const uint32_t t = h[i];
                if (t == 0 && !one_on_one) continue;
                if (t != last_t || !last_len) {
                    char* const b0 = w;
                    w = write_uint(w, t);
                    *w++ = '\n';
                    last_len = (size_t)(w - b0);
                    memcpy(last_line, b0, last_len);
                    last_t = t;
                } else {
                    memcpy(w, last_line, 16);      // (a line is at most 11 bytes; the buffer has the slack)
                    w += last_len;
                }
            }
        }
        out.append(scratch.p.get(), (size_t)(w - scratch.p.get()));
    };
    auto stage_b = [&](Job& J) {
        const char* d = J.d;
        std::vector<KmerPart>& parts = J.parts;
        const size_t np = J.np;
        size_t used = np;
        for (size_t i = 0; i < np; i++)
            if (parts[i].bad_utf8 || parts[i].panic) { used = i + 1; break; }
        const bool failed = parts[used - 1].bad_utf8 || parts[used - 1].panic;
        size_t n_rec = 0, n_frag = 0, n_bytes = 0;
        std::vector<size_t> rec_first(used + 1, 0);
        for (size_t i = 0; i < used; i++) {
            KmerPart& P = parts[i];
            rec_first[i] = n_rec;
            n_rec += P.hdr_off.size();
            P.rec_base = n_frag;                  // (first FRAGMENT of the part: emitted records only)
            P.byte_base = n_bytes;
            for (uint8_t e : P.emit) n_frag += e;
            n_bytes += P.n_bytes;
        }
        rec_first[used] = n_rec;
        size_t limit;
        if (failed) {
            // bad UTF-8: the error surfaces when the NEXT record is read, after n_rec good ones; panic: inside the last record
            const size_t e = parts[used - 1].panic ? g0 + n_rec - 1 : g0 + n_rec;
            limit = e / chunk * chunk;
        } else limit = (g0 + n_rec) / chunk * chunk;   // the open chunk at the end is held back (released at EOF below)
        double t0 = now();
        J.bytes.resize(n_bytes + 16);
        J.frag_off.resize(n_frag + 1);
        J.hits.resize(n_bytes + 16);
        J.frag_off[0] = 0;
        parallel_for(used, [&](size_t i) {
            const KmerPart& P = parts[i];
            if (P.n_bytes) memcpy(J.bytes.data() + P.byte_base, P.bytes.get(), P.n_bytes);
            uint64_t o = P.byte_base;
            size_t f = P.rec_base;
            for (size_t r = 0; r < P.emit.size(); r++)
                if (P.emit[r]) { o += P.seq_len[r]; J.frag_off[++f] = o; }
        });
        t_gather += now() - t0; t0 = now();
        if (n_frag) LIB(umgap_index_lookup_records(ix, J.bytes.data(), J.frag_off.data(), n_frag, 0u, J.hits.data()));
        t_lookup += now() - t0; t0 = now();
        if (J.tails.size() < used) J.tails.resize(used);
        J.latch.wait();                             // the writer thread is through with what this job printed two blocks ago
        parallel_for(used, [&](size_t i) {
            KmerPart& P = parts[i];
            const size_t nr = P.hdr_off.size(), lo = g0 + rec_first[i];
            const size_t cut = limit <= lo ? 0 : std::min(nr, limit - lo);
            P.out.clear();
            J.tails[i].clear();
            P.out.reserve(nr * 24 + P.n_bytes * 8);
            format_part(d, P, J.hits.data(), J.frag_off.data(), 0, cut, P.out);
            if (cut < nr && !failed) format_part(d, P, J.hits.data(), J.frag_off.data(), cut, nr, J.tails[i]);
        });
        t_format += now() - t0; t0 = now();
        const size_t held_from = g0 / chunk * chunk;
        if (!held.empty() && limit > held_from) { fwrite(held.data(), 1, held.size(), outf); held.clear(); }
        // (600 bytes per pair with -o: handed to the writer thread as they lie -- copying them into its 1 MB chunks was this stage's
        // largest single cost inside the script's chain)
        for (size_t i = 0; i < used; i++) write_owned_or_copy(outf, parts[i].out.data(), parts[i].out.size(), J.latch);
        t_write += now() - t0;
        if (failed) {
            fflush(outf);
            if (parts[used - 1].panic) throw CliPanic("byte index is not a char boundary");
            throw CliError("stream did not contain valid UTF-8");
        }
        for (size_t i = 0; i < used; i++) held.append(J.tails[i]);
        g0 += n_rec;
    };
    Job jobs[2];
    int cur = 0;
    std::thread helper;
    std::exception_ptr helper_err;
    auto join_helper = [&]() {
        if (helper.joinable()) helper.join();
        if (helper_err) { std::exception_ptr e = helper_err; helper_err = nullptr; std::rethrow_exception(e); }
    };
    try {
        const char* d;
        size_t len;
        for (;;) {
            double t0 = now();
            if (!blocks.next(d, len)) break;
            t_read += now() - t0; t0 = now();
            if (blocks.first_line_error()) { join_helper(); throw CliError("Expected > at beginning of fasta header."); }
            Job& J = jobs[cur];
            std::vector<size_t> cuts = split_records(d, len, T);
            J.np = cuts.size() - 1;
            J.d = d;
            if (J.parts.size() < J.np) J.parts.resize(J.np);
            parallel_for(J.np, [&](size_t i) { parse_kmer_part(d, cuts[i], cuts[i + 1], J.parts[i]); });
            t_parse += now() - t0; t0 = now();
            join_helper();                         // stage B of the previous block is done (and has thrown here if it failed)
            t_wait += now() - t0;
            bool failing = false;
            for (size_t i = 0; i < J.np; i++) failing = failing || J.parts[i].bad_utf8 || J.parts[i].panic;
            if (failing) { stage_b(J); continue; }
            helper = std::thread([&stage_b, &helper_err, &J]() { try { stage_b(J); } catch (...) { helper_err = std::current_exception(); } });
            cur ^= 1;
        }
        join_helper();
    } catch (...) {
        if (helper.joinable()) helper.join();
        throw;
    }
    fwrite(held.data(), 1, held.size(), outf);   // EOF closes the last chunk
    fflush(outf);
    if (timing) fprintf(stderr, "stream_kmer_records: stage A: read %.3f, parse %.3f, wait for stage B %.3f | stage B: gather %.3f, lookup %.3f, format %.3f, write %.3f\n",
                        t_read, t_parse, t_wait, t_gather, t_lookup, t_format, t_write);
}

// --shards N: the index sharded by bucket range over N GPUs -- what the resident `umgap prot2kmer2lca -m -o -s <socket>` server
// of scripts/umgap-analyse.sh:257-264 becomes when the index outgrows one GPU.  Rank r holds shard r (umgap_index_open_shard) on
// GPU r and looks up the r-th slice of every batch; umgap_sharded_lookup9 routes each k-mer to its owner and back.
//   * ranks as THREADS of this process (default): one host thread per GPU, the exchange over RCCL.
//   * ranks as PROCESSES (--shard-processes, or UMGAP_SHARDS_PROCS=1): this process is rank 0 and starts N - 1 workers
//     (`umgap __shard-worker`, started before anything here touches the GPU); the exchange runs over the IPC communicator
//     (umgap_comm_init_ipc: peer-written keys and values through hipIpc mappings -- between GPUs the same mappings travel over
//     xGMI), the batches reach the workers through one POSIX shared-memory block (control words, keys, taxa).
// UMGAP_SHARDS_LOCAL=1 puts every rank on GPU 0 (threads: the library's device-to-device hub; processes: the same IPC
// communicator) -- the rehearsal on a one-GPU box; the tests use it.
struct ShardProcCtl {                  // head of the shared block; keys and taxa follow at `keys_off` / `taxa_off`
    std::atomic<uint64_t> seq;         // rank 0 bumps it per command
    std::atomic<uint32_t> cmd;         // 1 = lookup of [lo[r], hi[r]), 2 = umgap_sharded_check, 3 = leave
    std::atomic<uint32_t> ready;       // workers whose group stands
    std::atomic<uint32_t> done;        // workers that finished command `seq`
    std::atomic<uint32_t> failed;      // a worker that could not come up (its message is in err[rank])
    uint32_t world;
    uint64_t max_keys, keys_off, taxa_off, bytes;
    uint64_t lo[64], hi[64];
    std::atomic<int32_t> rc[64];
    char err[64][232];
};
static_assert(std::atomic<uint64_t>::is_always_lock_free && std::atomic<uint32_t>::is_always_lock_free, "the control words live in shared memory");

inline void shard_proc_nap() { struct timespec ts{0, 50000}; nanosleep(&ts, nullptr); }

struct ShardGang {
    std::vector<umgap_index*> ix;
    std::vector<umgap_comm*> comm;
    std::vector<umgap_sharded*> sh;
    uint64_t max_keys = 48u << 20;     // per rank and call
    int n = 0;
    // ranks as processes
    bool procs = false;
    std::vector<pid_t> workers;        // ranks 1 .. n-1
    ShardProcCtl* ctl = nullptr;
    std::string shm_name;
    bool broken = false;               // a rank is gone: no collective call can complete any more
    template <class F>
    void each_rank(F f) {              // f(rank) on n threads; the first failure is rethrown
        std::vector<std::thread> ts;
        std::vector<std::exception_ptr> errs(n);
        for (int r = 0; r < n; r++) ts.emplace_back([&, r]() { try { f(r); } catch (...) { errs[r] = std::current_exception(); } });
        for (auto& t : ts) t.join();
        for (auto& e : errs) if (e) std::rethrow_exception(e);
    }
    // RCCL announces its version on stdout when the first communicator comes up; stdout is this command's data, so it
    // points at stderr until the gang stands
    struct StdoutToStderr {
        int saved;
        StdoutToStderr() { fflush(stdout); umgap_host::AsyncIo::drain_all(); saved = dup(1); if (saved >= 0) dup2(2, 1); }
        ~StdoutToStderr() { fflush(stdout); umgap_host::AsyncIo::drain_all(); if (saved >= 0) { dup2(saved, 1); close(saved); } }
    };
    ShardGang(const std::string& path, int shards, bool as_processes) : n(shards), procs(as_processes) {
        const bool local = getenv("UMGAP_SHARDS_LOCAL") != nullptr;
        if (shards < 1 || shards > 64) throw CliError("--shards must be in 1..64");
        if (procs) { start_processes(path, local); return; }
        if (!local && umgap_device_count() < shards) throw CliError("--shards " + std::to_string(shards) + " needs that many GPUs (" + std::to_string(umgap_device_count()) + " visible)");
        ix.assign(n, nullptr); comm.assign(n, nullptr); sh.assign(n, nullptr);
        uint8_t id[128];
        StdoutToStderr quiet;
        if (local) { if (umgap_comm_init_local(n, 0, comm.data()) != UMGAP_OK) throw CliError(umgap_last_error()); }
        else if (umgap_comm_unique_id(id) != UMGAP_OK) throw CliError(umgap_last_error());
        each_rank([&](int r) {
            const int dev = local ? 0 : r;
            LIB(umgap_index_open_shard(path.c_str(), 1u << dev, (uint32_t)r, (uint32_t)n, &ix[r]));
            if (umgap_index_kind(ix[r]) != 9) throw CliError("--shards needs a 9-mer index");
            if (!local) LIB(umgap_comm_init_rank(id, n, r, dev, &comm[r]));
            LIB(umgap_comm_selftest(comm[r], 1 << 16));   // a link that drops or garbles bytes shows up here, not in the answers
            LIB(umgap_sharded_create(ix[r], nullptr, comm[r], max_keys, 4, 0.05, &sh[r]));
        });
    }
    ~ShardGang() {
        if (procs) { stop_processes(); return; }
        for (auto s : sh) umgap_sharded_free(s);
        for (auto c : comm) umgap_comm_free(c);
        for (auto i : ix) umgap_index_free(i);
    }
    void lookup(const uint64_t* keys, size_t total, uint32_t* taxa) {
        if (procs) { lookup_processes(keys, total, taxa); return; }
        // every rank makes the same number of (collective) calls: rounds of at most n * max_keys keys, split evenly
        for (size_t done = 0; done < total;) {
            const size_t round = std::min<size_t>(total - done, (size_t)n * max_keys), per = (round + n - 1) / n;
            each_rank([&](int r) {
                const size_t lo = std::min(round, (size_t)r * per), hi = std::min(round, (size_t)(r + 1) * per);
                LIB(umgap_sharded_lookup9(sh[r], keys + done + lo, hi - lo, 0u, taxa + done + lo));
            });
            done += round;
        }
        // an overflowed exchange region would have answered some k-mers with a miss: refuse to print such a batch
        each_rank([&](int r) { LIB(umgap_sharded_check(sh[r])); });
    }

    // ---- ranks as processes ----------------------------------------------------------------------------------------------
    static void* map_block(const char* name, bool create, size_t bytes, std::string& why) {
        int fd = shm_open(name, create ? (O_CREAT | O_EXCL | O_RDWR) : O_RDWR, 0600);
        if (fd < 0) { why = std::string("shm_open ") + name + ": " + strerror(errno); return nullptr; }
        if (create && ftruncate(fd, (off_t)bytes) != 0) { why = std::string("ftruncate: ") + strerror(errno); close(fd); shm_unlink(name); return nullptr; }
        if (!create) { struct stat st; if (fstat(fd, &st) != 0 || (size_t)st.st_size < sizeof(ShardProcCtl)) { why = "shared block too small"; close(fd); return nullptr; } bytes = (size_t)st.st_size; }
        void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (p == MAP_FAILED) { why = std::string("mmap: ") + strerror(errno); if (create) shm_unlink(name); return nullptr; }
        return p;
    }
    uint64_t* shared_keys() const { return (uint64_t*)((char*)ctl + ctl->keys_off); }
    uint32_t* shared_taxa() const { return (uint32_t*)((char*)ctl + ctl->taxa_off); }
    // a worker that died leaves its collective partners waiting: say so instead of waiting with them
    void reap_or_throw() {
        for (size_t i = 0; i < workers.size(); i++) {
            if (workers[i] <= 0) continue;
            int st = 0;
            if (waitpid(workers[i], &st, WNOHANG) == workers[i]) {
                workers[i] = -1;
                broken = true;
                const int r = (int)i + 1;
                std::string msg = ctl && ctl->err[r][0] ? std::string(ctl->err[r]) : std::string("ended unexpectedly");
                throw CliError("--shards: rank " + std::to_string(r) + " " + msg);
            }
        }
    }
    void kill_workers() {
        for (auto& w : workers) if (w > 0) { kill(w, SIGTERM); int st; waitpid(w, &st, 0); w = -1; }
    }
    void start_processes(const std::string& path, bool local) {
        max_keys = 16u << 20;          // (the shared block holds world x max_keys keys and taxa: 192 MB per rank)
        if (const char* e = getenv("UMGAP_SHARDS_MAX_KEYS")) { uint64_t v = strtoull(e, nullptr, 10); if (v >= 1024 && v <= (1ull << 28)) max_keys = v; }
        char tag[64];
        snprintf(tag, sizeof tag, "%d-%llx", (int)getpid(), (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count());
        shm_name = std::string("/umgap-gang-") + tag;
        const std::string ipc_name = std::string("/umgap-hub-") + tag;
        const size_t head = (sizeof(ShardProcCtl) + 4095) & ~(size_t)4095;
        const size_t bytes = head + (size_t)n * max_keys * 12;
        std::string why;
        ctl = (ShardProcCtl*)map_block(shm_name.c_str(), true, bytes, why);
        if (!ctl) throw CliError("--shards: " + why);
        new (ctl) ShardProcCtl();      // (a fresh shared-memory object is zero-filled; this names the atomics' lifetime)
        ctl->world = (uint32_t)n; ctl->max_keys = max_keys; ctl->keys_off = head; ctl->taxa_off = head + (size_t)n * max_keys * 8; ctl->bytes = bytes;
        ix.assign(1, nullptr); comm.assign(1, nullptr); sh.assign(1, nullptr);
        try {
            // the workers first: nothing in this process has touched the GPU yet
            for (int r = 1; r < n; r++) {
                const std::string rs = std::to_string(r), ws = std::to_string(n), ds = std::to_string(local ? 0 : r), ps = std::to_string((long)getpid());
                const char* argv[] = {"umgap", "__shard-worker", path.c_str(), rs.c_str(), ws.c_str(), ds.c_str(), ipc_name.c_str(), shm_name.c_str(), ps.c_str(), nullptr};
                posix_spawn_file_actions_t fa;
                posix_spawn_file_actions_init(&fa);
                posix_spawn_file_actions_addopen(&fa, 0, "/dev/null", O_RDONLY, 0);
                posix_spawn_file_actions_adddup2(&fa, 2, 1);      // a worker's stdout is not this command's data
                pid_t pid = 0;
                const int e = posix_spawn(&pid, "/proc/self/exe", &fa, nullptr, (char* const*)argv, environ);
                posix_spawn_file_actions_destroy(&fa);
                if (e != 0) throw CliError(std::string("--shards: cannot start a worker: ") + strerror(e));
                workers.push_back(pid);
            }
            if (!local && umgap_device_count() < n) throw CliError("--shards " + std::to_string(n) + " needs that many GPUs (" + std::to_string(umgap_device_count()) + " visible)");
            StdoutToStderr quiet;
            LIB(umgap_index_open_shard(path.c_str(), 1u, 0u, (uint32_t)n, &ix[0]));
            if (umgap_index_kind(ix[0]) != 9) throw CliError("--shards needs a 9-mer index");
            LIB(umgap_comm_init_ipc(ipc_name.c_str(), n, 0, 0, &comm[0]));
            LIB(umgap_comm_selftest(comm[0], 1 << 16));
            LIB(umgap_sharded_create(ix[0], nullptr, comm[0], max_keys, 4, 0.05, &sh[0]));
            while (ctl->ready.load(std::memory_order_acquire) < (uint32_t)(n - 1)) { reap_or_throw(); shard_proc_nap(); }
        } catch (...) {
            std::string first;            // a worker's own message says more than the time-out it caused here
            if (ctl->failed.load()) for (int r = 1; r < n && first.empty(); r++) if (ctl->err[r][0]) first = "--shards: rank " + std::to_string(r) + " " + ctl->err[r];
            kill_workers();
            shm_unlink(shm_name.c_str());
            munmap(ctl, bytes); ctl = nullptr;
            if (!first.empty()) throw CliError(first);
            throw;
        }
        shm_unlink(shm_name.c_str());     // everyone is attached: the block goes with its last mapping
    }
    void command(uint32_t cmd) {
        ctl->done.store(0, std::memory_order_relaxed);
        ctl->cmd.store(cmd, std::memory_order_relaxed);
        ctl->seq.fetch_add(1, std::memory_order_release);
    }
    void wait_done() {
        while (ctl->done.load(std::memory_order_acquire) < (uint32_t)(n - 1)) { reap_or_throw(); shard_proc_nap(); }
        for (int r = 1; r < n; r++) if (ctl->rc[r].load()) throw CliError(std::string(ctl->err[r]));
    }
    void lookup_processes(const uint64_t* keys, size_t total, uint32_t* taxa) {
        if (broken) throw CliError("--shards: the gang has lost a rank");      // (a resident server's later connections end here)
        for (size_t done = 0; done < total;) {
            const size_t round = std::min<size_t>(total - done, (size_t)n * max_keys), per = (round + n - 1) / n;
            for (int r = 0; r < n; r++) { ctl->lo[r] = std::min(round, (size_t)r * per); ctl->hi[r] = std::min(round, (size_t)(r + 1) * per); }
            if (round > ctl->hi[0]) memcpy(shared_keys() + ctl->hi[0], keys + done + ctl->hi[0], (round - ctl->hi[0]) * 8);   // (rank 0 reads its slice where it lies)
            command(1);
            const int rc0 = umgap_sharded_lookup9(sh[0], keys + done, ctl->hi[0], 0u, taxa + done);
            const std::string err0 = rc0 ? umgap_last_error() : "";
            wait_done();               // (the call is collective: the others are answered whether or not this rank failed)
            if (rc0) throw CliError(err0);
            if (round > ctl->hi[0]) memcpy(taxa + done + ctl->hi[0], shared_taxa() + ctl->hi[0], (round - ctl->hi[0]) * 4);
            done += round;
        }
        command(2);
        const int rc0 = umgap_sharded_check(sh[0]);
        const std::string err0 = rc0 ? umgap_last_error() : "";
        wait_done();
        if (rc0) throw CliError(err0);
    }
    void stop_processes() {
        if (ctl) {
            if (broken) kill_workers();     // (the group's collective free would wait for the rank that is gone: the process ends instead)
            else {
                command(3);            // (umgap_sharded_free is collective in put mode: every rank leaves together)
                umgap_sharded_free(sh[0]); umgap_comm_free(comm[0]); umgap_index_free(ix[0]);
                for (auto& w : workers) if (w > 0) { int st; waitpid(w, &st, 0); w = -1; }
            }
            munmap(ctl, ctl->bytes);
            ctl = nullptr;
        }
    }
};

// `umgap __shard-worker <index> <rank> <world> <device> <hub name> <block name> <parent pid>`: rank >= 1 of a --shards gang
// whose ranks are processes (started by ShardGang::start_processes, not by hand).
int cmd_shard_worker(const std::vector<std::string>& a) {
    if (a.size() != 7) usage_error("__shard-worker is started by `umgap prot2kmer2lca --shards N --shard-processes`");
    const std::string& path = a[0];
    const int rank = atoi(a[1].c_str()), world = atoi(a[2].c_str()), dev = atoi(a[3].c_str());
    const pid_t parent = (pid_t)atol(a[6].c_str());
    prctl(PR_SET_PDEATHSIG, SIGTERM);  // rank 0 gone => so is the gang
    if (world < 2 || world > 64 || rank < 1 || rank >= world || dev < 0 || dev > 31) usage_error("__shard-worker: bad rank / world / device");
    std::string why;
    ShardProcCtl* ctl = (ShardProcCtl*)ShardGang::map_block(a[5].c_str(), false, 0, why);
    if (!ctl) throw CliError(why);
    auto note = [&](const std::string& m) { snprintf(ctl->err[rank], sizeof ctl->err[rank], "%s", m.c_str()); };
    umgap_index* ix = nullptr; umgap_comm* comm = nullptr; umgap_sharded* sh = nullptr;
    auto up = [&]() -> bool {
        if (getppid() != parent) { note("lost its parent"); return false; }
        if (umgap_index_open_shard(path.c_str(), 1u << dev, (uint32_t)rank, (uint32_t)world, &ix) != UMGAP_OK) { note(umgap_last_error()); return false; }
        if (umgap_index_kind(ix) != 9) { note("--shards needs a 9-mer index"); return false; }
        if (umgap_comm_init_ipc(a[4].c_str(), world, rank, dev, &comm) != UMGAP_OK) { note(umgap_last_error()); return false; }
        if (umgap_comm_selftest(comm, 1 << 16) != UMGAP_OK) { note(umgap_last_error()); return false; }
        if (umgap_sharded_create(ix, nullptr, comm, ctl->max_keys, 4, 0.05, &sh) != UMGAP_OK) { note(umgap_last_error()); return false; }
        return true;
    };
    if (!up()) { ctl->failed.fetch_add(1); fprintf(stderr, "umgap shard worker %d: %s\n", rank, ctl->err[rank]); return 1; }
    ctl->ready.fetch_add(1, std::memory_order_release);
    uint64_t* keys = (uint64_t*)((char*)ctl + ctl->keys_off);
    uint32_t* taxa = (uint32_t*)((char*)ctl + ctl->taxa_off);
    uint64_t seen = 0;
    for (;;) {
        uint64_t s;
        while ((s = ctl->seq.load(std::memory_order_acquire)) == seen) {
            if (getppid() != parent) return 1;
            shard_proc_nap();
        }
        seen = s;
        const uint32_t cmd = ctl->cmd.load(std::memory_order_relaxed);
        if (cmd == 3) break;
        int rc = UMGAP_OK;
        if (cmd == 1) rc = umgap_sharded_lookup9(sh, keys + ctl->lo[rank], ctl->hi[rank] - ctl->lo[rank], 0u, taxa + ctl->lo[rank]);
        else if (cmd == 2) rc = umgap_sharded_check(sh);
        if (rc) note(umgap_last_error());
        ctl->rc[rank].store(rc);
        ctl->done.fetch_add(1, std::memory_order_release);
    }
    umgap_sharded_free(sh); umgap_comm_free(comm); umgap_index_free(ix);
    return 0;
}

umgap_index* open_index(const std::string& path) {
    umgap_index* ix = nullptr;
    int rc = umgap_index_open(path.c_str(), 1u, &ix);
    if (rc) lib_fail(rc);
    return ix;
}

// record loop shared by pept2lca (variable-length index) and prot2tryp2lca: `expand` turns one
// sequence line into the peptides to look up
// `unwrap`: the record's lines are one sequence (prot2kmer2lca reads with unwrap = true, src/io/fasta.rs:62-64).
// `keep_record` (optional): a record it rejects prints nothing, not even its header (src/commands/prot2kmer2lca.rs:172),
// but still counts towards its chunk.
void stream_strings(FILE* in, FILE* outf, umgap_index* ix, bool one_on_one, size_t chunk,
                    const std::function<void(const std::string&, StringBatch&)>& expand, bool unwrap = false,
                    const std::function<bool(const Record&)>& keep_record = nullptr) {
    if (chunk == 0) chunk = 1;
    FastaReader reader(in, unwrap);
    Out out(outf);
    std::vector<std::string> headers;
    std::vector<char> skip;     // record prints nothing at all
    std::vector<size_t> first;  // index of the record's first peptide in the batch
    StringBatch batch;
    std::vector<uint32_t> taxa;
    // `keep` = number of leading records of the batch to print (batches start on chunk boundaries)
    auto run = [&](size_t keep) {
        lookup_strings(ix, batch, taxa);
        for (size_t r = 0; r < headers.size() && r < keep; r++) {
            if (skip[r]) continue;
            out.buf.push_back('>'); out.buf.append(headers[r]); out.buf.push_back('\n');
            size_t end = r + 1 < headers.size() ? first[r + 1] : batch.size();
            for (size_t i = first[r]; i < end; i++)
                if (taxa[i] != 0 || one_on_one) { append_u64(out.buf, taxa[i]); out.buf.push_back('\n'); }
            out.maybe_flush();
        }
        headers.clear(); first.clear(); batch.clear(); skip.clear();
    };
    Record rec;
    try {
        while (reader.read_record(rec)) {
            first.push_back(batch.size());
            const bool kept = !keep_record || keep_record(rec);
            skip.push_back(kept ? 0 : 1);
            headers.push_back(std::move(rec.header));
            if (kept) for (auto& s : rec.sequence) expand(s, batch);
            // flush on chunk boundaries only, so that an error can still drop the whole open chunk
            if ((batch.bytes.size() >= (64u << 20) || headers.size() >= (1u << 20)) && headers.size() % chunk == 0) run(headers.size());
        }
    } catch (CliError&) {
        run(headers.size() / chunk * chunk);   // ChunkedRecords drops the chunk the error falls in (src/io/fasta.rs:125-131)
        out.flush();
        throw;
    } catch (CliPanic&) {
        run((headers.size() - 1) / chunk * chunk);   // a panic inside the record just pushed: its chunk is lost as well
        out.flush();
        throw;
    }
    run(headers.size());
    out.flush();
}

int cmd_pept2lca(const std::vector<std::string>& args) {
    Parsed p = parse_args(args, {{'o', "one-on-one", false}, {'m', "in-memory", false}, {'c', "chunksize", true}});
    size_t chunk = (size_t)opt_usize(p, "chunksize", 240);  // decides what is printed before an error (see stream_lookup)
    std::string path = need_positional(p, "fst-file");
    umgap_index* ix = open_index(path);  // -m: the table is always resident (in HBM)
    if (umgap_index_kind(ix) == 2) {
        const LineExpand whole = [](std::string_view line, StringBatch& b) { b.add(line); };
        stream_lookup(stdin, stdout, LookupFn(), p.has("one-on-one"), false, 9, chunk, &whole, ix);
    } else stream_lookup(stdin, stdout, ix, p.has("one-on-one"), false, 9, chunk);
    umgap_index_free(ix);
    return 0;
}

// prot2tryp2lca: src/commands/prot2tryp2lca.rs:88-140.  The default pattern "([KR])([^P])" applied twice
// and '*' -> newline is: cut after every K/R that is not followed by P; '*' separates and is dropped.
int cmd_prot2tryp2lca(const std::vector<std::string>& args) {
    Parsed p = parse_args(args, {{'o', "one-on-one", false}, {'m', "in-memory", false}, {'c', "chunksize", true}, {'p', "pattern", true},
                                 {'l', "minlen", true}, {'L', "maxlen", true}, {'k', "keep", true}, {'d', "drop", true}});
    size_t chunk = (size_t)opt_usize(p, "chunksize", 240);
    size_t minlen = (size_t)opt_usize(p, "minlen", 5), maxlen = (size_t)opt_usize(p, "maxlen", 50);
    std::string keep = p.get("keep").value_or(""), drop = p.get("drop").value_or("");
    const std::string pattern = p.get("pattern").value_or("([KR])([^P])");
    const bool default_pattern = pattern == "([KR])([^P])";
    // -p <regex> (src/commands/prot2tryp2lca.rs:65-68,96,112-115): replace_all(pattern, "$1\n$2") applied twice, host side.
    // The reference compiles it with Rust's `regex` crate; cleave_pattern.hpp rewrites that syntax for std::regex and refuses
    // what the crate refuses (look-around, backreferences).  The default pattern keeps the scanner below.
    CleavePattern custom;
    if (!default_pattern) {
        try { custom = compile_cleave_pattern(pattern); }
        catch (const CleavePatternError& e) { throw CliError(e.what()); }
    }
    std::string path = need_positional(p, "fst-file");
    umgap_index* ix = open_index(path);
    auto expand = [&](std::string_view seq_in, StringBatch& b) {
        std::string cut;
        if (!default_pattern) {
            cut = custom.cut(std::string(seq_in));
        }
        const std::string_view seq = default_pattern ? seq_in : std::string_view(cut);
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
            else if (default_pattern && (c == 'K' || c == 'R') && i + 1 < seq.size() && seq[i + 1] != 'P') { emit(st, i + 1); st = i + 1; }
        }
        emit(st, seq.size());
    };
    // a byte-string index (what tryptic peptides live in): block-parallel, one look-up per block; a packed 9-mer table: the
    // record-at-a-time loop (only peptides of nine residues could hit there)
    if (umgap_index_kind(ix) == 2) {
        const LineExpand per_line = expand;
        stream_lookup(stdin, stdout, LookupFn(), p.has("one-on-one"), false, 9, chunk, &per_line, ix);
    } else stream_strings(stdin, stdout, ix, p.has("one-on-one"), chunk, [&](const std::string& s, StringBatch& b) { expand(s, b); });
    umgap_index_free(ix);
    return 0;
}

int cmd_prot2kmer2lca(const std::vector<std::string>& args) {
    Parsed p = parse_args(args, {{'k', "length", true}, {'o', "one-on-one", false}, {'s', "socket", true}, {'m', "in-memory", false}, {'c', "chunksize", true},
                                 {0, "shards", true}, {0, "shard-processes", false}});
    size_t k = (size_t)opt_usize(p, "length", 9);
    size_t chunk = (size_t)opt_usize(p, "chunksize", 240);
    std::string path = need_positional(p, "fst-file");
    bool o = p.has("one-on-one");
    std::unique_ptr<ShardGang> gang;
    if (p.has("shards")) gang.reset(new ShardGang(path, (int)opt_usize(p, "shards", 1), p.has("shard-processes") || getenv("UMGAP_SHARDS_PROCS") != nullptr));
    umgap_index* ix = gang ? nullptr : open_index(path);
    // A packed 9-mer table answers k = 9; any other k can only miss there (its keys are 9 residues long), which the fast
    // path emits without asking the device.  A byte-string (variable-length) table serves EVERY k like the reference's fst
    // does (src/commands/prot2kmer2lca.rs:71-73,172-179): all windows of k residues through umgap_index_lookup_var.
    const bool var_index = !gang && umgap_index_kind(ix) == 2;
    auto stream = [&](FILE* in, FILE* outf) {
        if (gang) { stream_lookup(in, outf, [&](const uint64_t* keys, size_t n, uint32_t* taxa) { gang->lookup(keys, n, taxa); }, o, true, k, chunk); return; }
        if (!var_index && k == 9 && !getenv("UMGAP_KMER_KEYS")) { stream_kmer_records(in, outf, ix, o, chunk); return; }   // (UMGAP_KMER_KEYS=1: keys packed on the host, A/B)
        if (!var_index) { stream_lookup(in, outf, ix, o, true, k, chunk); return; }
        if (k >= 1) {      // block-parallel, the windows of k residues as strings (parse_lookup_part)
            const LineExpand windows = [](std::string_view, StringBatch&) {};   // (its presence selects the string mode; k-mer mode cuts the windows itself)
            stream_lookup(in, outf, LookupFn(), o, true, k, chunk, &windows, ix);
            return;
        }
        stream_strings(in, outf, ix, o, chunk,
                       [&](const std::string& seq, StringBatch& b) {
                           for (size_t i = 0; i + k <= seq.size(); i++) {
                               // &prot[i..i + k] panics unless both ends are char boundaries
                               if (((unsigned char)seq[i] & 0xC0) == 0x80 || (i + k < seq.size() && ((unsigned char)seq[i + k] & 0xC0) == 0x80))
                                   throw CliPanic("byte index is not a char boundary");
                               b.add(std::string_view(seq.data() + i, k));
                           }
                       },
                       true, [&](const Record& r) { return !r.sequence.empty() && r.sequence[0].size() >= k; });
    };
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
            const bool async_io = umgap_host::AsyncIo::enabled();
            FILE* in = async_io ? umgap_host::AsyncIo::reader(c, true) : fdopen(c, "rb");
            FILE* outf = async_io ? umgap_host::AsyncIo::writer(dup(c), true) : fdopen(dup(c), "wb");
            try {
                stream(in, outf);
                printf("Connection finished succesfully.\n");
            } catch (CliError& e) { printf("Connection died with an error: %s\n", e.what()); }
            fflush(stdout);
            fclose(outf);
            fclose(in);
        }
    }
    stream(stdin, stdout);
    umgap_index_free(ix);
    return 0;
}

// ---- taxa2agg ------------------------------------------------------------------------------------
umgap_taxonomy* load_taxonomy(const std::string& path) {
    const bool timing = getenv("UMGAP_TIMING") != nullptr;
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    TaxonRows rows = read_taxa_file_parallel(path);
    const double t1 = now();
    std::vector<uint32_t> id(rows.id.size()), parent(rows.id.size());
    for (size_t i = 0; i < rows.id.size(); i++) {
        if (rows.id[i] >= 0xFFFFFFFFull || rows.parent[i] >= 0xFFFFFFFFull) throw CliError("taxon id >= 2^32-1 is not supported");
        id[i] = (uint32_t)rows.id[i];
        parent[i] = (uint32_t)rows.parent[i];
    }
    umgap_taxonomy* tx = nullptr;
    int rc = umgap_taxonomy_create(id.data(), parent.data(), rows.valid.data(), rows.rank.data(), id.size(), 0, &tx);
    if (timing) fprintf(stderr, "taxonomy: parse %.3f s, create (host tables + upload, includes HIP start-up when first) %.3f s\n", t1 - t0, now() - t1);
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

    // block-parallel record loop (fastpath.hpp), two overlapped stages: while block n is aggregated on the device, formatted and
    // written (stage B, a helper thread), block n + 1 is read and tokenized (stage A).  Stage B runs one block at a time, in input
    // order, so output and errors come as from the plain loop of src/commands/taxa2agg.rs:160-180.
    struct alignas(128) AggPart {
        std::vector<size_t> hdr_off;
        std::vector<uint32_t> hdr_len;
        std::vector<uint64_t> offsets;     // n_records + 1, relative to the part's first taxon
        std::vector<uint32_t> taxa;
        std::vector<float> scores;
        std::vector<uint64_t> big;         // ids that do not fit 32 bits, for the message
        std::optional<CliError> error;     // stops the stream after the records parsed so far
        size_t rec_base = 0, tax_base = 0;
        std::string out;
    };
    auto parse_part = [&](const char* d, size_t begin, size_t end, AggPart& P) {
        // (the vectors are filled on this thread's stack and handed back: see LookupPart)
        std::vector<size_t> hdr_off = std::move(P.hdr_off);
        std::vector<uint32_t> hdr_len = std::move(P.hdr_len), taxa = std::move(P.taxa);
        std::vector<uint64_t> offsets = std::move(P.offsets), big = std::move(P.big);
        std::vector<float> scores = std::move(P.scores);
        hdr_off.clear(); hdr_len.clear(); taxa.clear(); offsets.clear(); big.clear(); scores.clear();
        offsets.push_back(0);
        taxa.reserve((end - begin) / 4 + 16);
        P.error.reset();
        P.out.clear();
        bool open = false, stop = false;
        auto close = [&]() { if (open) { offsets.push_back(taxa.size()); open = false; } };
        auto line = [&](size_t s, size_t e, bool high) {
            size_t le = e;
            if (le > s && d[le - 1] == '\r') le--;
            if (high && !valid_utf8(std::string_view(d + s, le - s))) {   // the record read so far still counts, then the stream errors
                close();
                P.error = CliError("stream did not contain valid UTF-8");
                stop = true;
                return;
            }
            if (le > s && d[s] == '>') {
                close();
                hdr_off.push_back(s + 1);
                hdr_len.push_back((uint32_t)(le - s - 1));
                open = true;
                return;
            }
            try {
                uint64_t t = 0;
                const size_t n = le - s;
                bool fast = !scored && n >= 1 && n <= 18;     // the common line: a few decimal digits
                for (size_t i = 0; fast && i < n; i++) {
                    const unsigned c = (unsigned char)d[s + i] - '0';
                    if (c > 9) fast = false; else t = t * 10 + c;
                }
                if (!fast) {
                    const std::string_view l(d + s, n);
                    if (scored) {  // with_score, src/commands/taxa2agg.rs:142-148
                        const size_t eq = l.find('=');
                        if (eq == std::string_view::npos || l.find('=', eq + 1) != std::string_view::npos) throw CliError("Taxon without score");
                        t = parse_usize(l.substr(0, eq));
                        scores.push_back(parse_f32(l.substr(eq + 1)));
                    } else t = parse_usize(l);
                }
                if (t >= 0xFFFFFFFEull) { big.push_back(t); t = 0xFFFFFFFEull; }
                taxa.push_back((uint32_t)t);
            } catch (CliError& err) {
                // the failing record is dropped; everything before it is still aggregated and printed
                P.error = err;
                hdr_off.pop_back();
                hdr_len.pop_back();
                taxa.resize(offsets.back());
                if (scored) scores.resize(offsets.back());
                open = false;
                stop = true;
            }
        };
        // the common line -- a few decimal digits and '\n' -- is read where it stands, value and end of line in one pass; every other
        // line (headers, scores, signs, CRs, high bytes, the unterminated last line) goes through line()
        size_t pos = begin;
        while (pos < end && !stop) {
            if (!scored) {
                while (pos < end) {
                    size_t q = pos;
                    uint64_t v = 0;
                    unsigned c;
                    while (q < end && (c = (unsigned)(unsigned char)d[q] - '0') <= 9) { v = v * 10 + c; q++; }
                    if (q == pos || q - pos > 9 || q >= end || d[q] != '\n') break;
                    taxa.push_back((uint32_t)v);
                    pos = q + 1;
                }
                if (pos >= end) break;
            }
            const char* nl = (const char*)memchr(d + pos, '\n', end - pos);
            const size_t e = nl ? (size_t)(nl - d) : end;
            bool high = false;
            for (size_t i = pos; i < e; i++) if ((unsigned char)d[i] >= 0x80) { high = true; break; }
            line(pos, e, high);
            pos = nl ? e + 1 : end;
        }
        if (!stop) close();
        P.hdr_off = std::move(hdr_off); P.hdr_len = std::move(hdr_len); P.taxa = std::move(taxa);
        P.offsets = std::move(offsets); P.big = std::move(big); P.scores = std::move(scores);
    };
    struct Job {
        const char* d = nullptr;
        std::vector<AggPart> parts;
        size_t np = 0;
        PinnedVec<uint32_t> taxa, out_taxon, out_status;
        PinnedVec<float> scores;
        PinnedVec<uint64_t> offsets;
    };
    BlockReader blocks(stdin);
    blocks.keep_previous_block();
    blocks.map_regular_file();
    const unsigned T = worker_count();
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const bool timing = getenv("UMGAP_TIMING") != nullptr;
    double t_read = 0, t_parse = 0, t_wait = 0, t_gather = 0, t_dev = 0, t_format = 0, t_write = 0;
    auto stage_b = [&](Job& J) {
        const char* d = J.d;
        std::vector<AggPart>& parts = J.parts;
        const size_t np = J.np;
        size_t used = np;
        for (size_t i = 0; i < np; i++)
            if (parts[i].error) { used = i + 1; break; }
        size_t n_rec = 0, n_tax = 0;
        for (size_t i = 0; i < used; i++) { parts[i].rec_base = n_rec; parts[i].tax_base = n_tax; n_rec += parts[i].hdr_off.size(); n_tax += parts[i].taxa.size(); }
        double t0 = now();
        J.taxa.resize(n_tax + 1);
        if (scored) J.scores.resize(n_tax + 1);
        J.offsets.resize(n_rec + 1);
        J.out_taxon.resize(n_rec + 1);
        J.out_status.resize(n_rec + 1);
        J.offsets[0] = 0;
        parallel_for(used, [&](size_t i) {
            const AggPart& P = parts[i];
            if (!P.taxa.empty()) memcpy(J.taxa.data() + P.tax_base, P.taxa.data(), P.taxa.size() * 4);
            if (scored && !P.scores.empty()) memcpy(J.scores.data() + P.tax_base, P.scores.data(), P.scores.size() * 4);
            for (size_t r = 0; r < P.hdr_off.size(); r++) J.offsets[P.rec_base + r + 1] = P.tax_base + P.offsets[r + 1];
        });
        t_gather += now() - t0; t0 = now();
        if (n_rec) LIB(umgap_aggregate(tx, method, strategy, factor, lower_bound, ranked ? 1 : 0, J.taxa.data(), scored ? J.scores.data() : nullptr,
                                       J.offsets.data(), n_rec, J.out_taxon.data(), J.out_status.data()));
        t_dev += now() - t0; t0 = now();
        // the first failing record (if any) ends the output
        size_t bad = n_rec;
        for (size_t r = 0; r < n_rec; r++)
            if (J.out_status[r] == UMGAP_READ_UNKNOWN_TAXON || J.out_status[r] == UMGAP_READ_PANIC || J.out_status[r] == 4u) { bad = r; break; }
        parallel_for(used, [&](size_t pi) {
            AggPart& P = parts[pi];
            P.out.reserve(P.hdr_off.size() * 32);
            for (size_t r = 0; r < P.hdr_off.size() && P.rec_base + r < bad; r++) {
                P.out.push_back('>');
                P.out.append(d + P.hdr_off[r], P.hdr_len[r]);
                P.out.push_back('\n');
                append_uint(P.out, J.out_taxon[P.rec_base + r]);
                P.out.push_back('\n');
            }
        });
        t_format += now() - t0; t0 = now();
        for (size_t i = 0; i < used; i++) fwrite(parts[i].out.data(), 1, parts[i].out.size(), stdout);
        t_write += now() - t0;
        if (bad < n_rec) {
            fflush(stdout);
            if (J.out_status[bad] == UMGAP_READ_UNKNOWN_TAXON) {
                uint64_t t = J.out_taxon[bad];
                if (t == 0xFFFFFFFEu)
                    for (size_t i = 0; i < used; i++) if (!parts[i].big.empty()) { t = parts[i].big[0]; break; }
                throw CliError("Unknown Taxon ID: " + std::to_string(t));
            }
            if (J.out_status[bad] == 4u) throw CliError("a read holds more than 1024 distinct taxa: not supported by the device aggregator");
            throw CliPanic("index out of bounds");
        }
        if (used < np || parts[used - 1].error) { fflush(stdout); throw *parts[used - 1].error; }
    };
    Job jobs[2];
    int cur = 0;
    std::thread helper;
    std::exception_ptr helper_err;
    auto join_helper = [&]() {
        if (helper.joinable()) helper.join();
        if (helper_err) { std::exception_ptr e = helper_err; helper_err = nullptr; std::rethrow_exception(e); }
    };
    try {
        const char* d;
        size_t len;
        for (;;) {
            double t0 = now();
            if (!blocks.next(d, len)) break;
            t_read += now() - t0; t0 = now();
            if (blocks.first_line_error()) { join_helper(); throw CliError("Expected > at beginning of fasta header."); }
            Job& J = jobs[cur];
            std::vector<size_t> cuts = split_records(d, len, T);
            J.np = cuts.size() - 1;
            J.d = d;
            if (J.parts.size() < J.np) J.parts.resize(J.np);
            parallel_for(J.np, [&](size_t pi) { parse_part(d, cuts[pi], cuts[pi + 1], J.parts[pi]); });
            t_parse += now() - t0; t0 = now();
            join_helper();                         // stage B of the previous block is done (and has thrown here if it failed)
            t_wait += now() - t0;
            bool failing = false;
            for (size_t i = 0; i < J.np; i++) failing = failing || J.parts[i].error.has_value();
            if (failing) { stage_b(J); continue; }  // the stream ends in this block
            helper = std::thread([&stage_b, &helper_err, &J]() { try { stage_b(J); } catch (...) { helper_err = std::current_exception(); } });
            cur ^= 1;
        }
        join_helper();
    } catch (...) {
        if (helper.joinable()) helper.join();
        throw;
    }
    if (timing) fprintf(stderr, "taxa2agg: stage A: read %.3f, parse %.3f, wait for stage B %.3f | stage B: gather %.3f, aggregate %.3f, format %.3f, write %.3f\n",
                        t_read, t_parse, t_wait, t_gather, t_dev, t_format, t_write);
    fflush(stdout);
    umgap_taxonomy_free(tx);
    return 0;
}

// ---- seedextend: the text filter (the fused pipeline runs it on the device) ---------------------------
// -r <taxons.tsv>: TaxonList::score (src/taxon.rs:181-191) of a taxon = Rank::score (src/rank.rs:86-99) of its nearest
// ancestor-or-self that is ranked or the root.  `self < &Rank::Species` is the first arm of Rank::score and PartialOrd is
// None for NoRank, so every named rank above species scores 12 and everything else is None (=> the -p penalty).
struct RankScores {
    std::vector<int8_t> present, ranked_or_root;   // by id
    std::vector<uint64_t> parent;
    std::vector<uint8_t> rank;
    std::vector<int16_t> memo;                     // -2 not computed, -1 None, else the score
    explicit RankScores(const TaxonRows& rows) {
        uint64_t max_id = 0;
        for (uint64_t id : rows.id) max_id = std::max(max_id, id);
        size_t len = (size_t)max_id + 1;            // src/taxon.rs:136-145
        present.assign(len, 0); parent.assign(len, 0); rank.assign(len, 0); memo.assign(len, -2);
        for (size_t i = 0; i < rows.id.size(); i++) { size_t id = (size_t)rows.id[i]; present[id] = 1; parent[id] = rows.parent[i]; rank[id] = rows.rank[i]; }
        if (!present[0]) { present[0] = 1; parent[0] = 0; rank[0] = 0; }   // new_with_unknown(.., true), :149-155
    }
    // None = -1
    int score(uint64_t index) {
        uint64_t cur = index;
        std::vector<uint64_t> path;
        int res = -1;
        for (size_t steps = 0;; steps++) {
            if (cur >= present.size() || !present[cur]) { res = -1; break; }        // get() == None ends the walk
            if (memo[cur] != -2) { res = memo[cur]; break; }
            if (parent[cur] == cur || rank[cur] != 0) { res = (rank[cur] != 0 && rank[cur] < 27) ? 12 : -1; memo[cur] = (int16_t)res; break; }
            if (steps > present.size()) throw CliError("taxonomy has a parent cycle without a ranked taxon");   // the reference would not terminate
            path.push_back(cur);
            cur = parent[cur];
        }
        for (uint64_t v : path) memo[v] = (int16_t)res;
        return res;
    }
};

int cmd_seedextend(const std::vector<std::string>& args) {
    Parsed p = parse_args(args, {{'s', "min-seed-size", true}, {'g', "max-gap-size", true}, {'r', "ranked", true}, {'p', "penalty", true}});
    uint64_t s = opt_usize(p, "min-seed-size", 2), g = opt_usize(p, "max-gap-size", 0);
    uint64_t penalty = opt_usize(p, "penalty", 5);
    if (!p.positional.empty()) usage_error("Found argument '" + p.positional[0] + "' which wasn't expected, or isn't valid in this context");
    std::optional<RankScores> ranks;
    if (auto tf = p.get("ranked")) {
        RankScores rs(read_taxa_file_parallel(*tf));
        for (size_t id = 0; id < rs.present.size(); id++) (void)rs.score(id);   // memoise everything: workers only read
        ranks.emplace(std::move(rs));
    }
    auto score_of = [&](uint64_t t) -> uint64_t {
        if (t >= ranks->memo.size() || ranks->memo[t] < 0) return penalty;
        return (uint64_t)ranks->memo[t];
    };
    // host restatement of the record state machine for the text filter (the device kernels
    // umgap_seedextend_dev / umgap_classify9_dev serve the fused pipeline), block-parallel over
    // records.  src/commands/seedextend.rs:99-148
    struct alignas(128) SePart { std::string out; std::optional<CliError> error; bool panic = false; };
    BlockReader blocks(stdin);
    const unsigned T = worker_count();
    const char* d;
    size_t len;
    blocks.map_regular_file();
    // (kept from block to block: the output strings keep their memory; two sets take turns, because a set's strings belong to the
    // writer thread until it has written them -- they leave without a copy)
    std::vector<SePart> part_sets[2];
    OutLatch latches[2];
    for (int turn = 0; blocks.next(d, len); turn ^= 1) {
        if (blocks.first_line_error()) throw CliError("Expected > at beginning of fasta header.");
        std::vector<size_t> cuts = split_records(d, len, T);
        const size_t np = cuts.size() - 1;
        std::vector<SePart>& parts = part_sets[turn];
        OutLatch& latch = latches[turn];
        latch.wait();
        if (parts.size() < np) parts.resize(np);
        parallel_for(np, [&](size_t pi) {
            SePart& P = parts[pi];
            P.out.clear();
            P.error.reset();
            P.panic = false;
            P.out.reserve((cuts[pi + 1] - cuts[pi]) + 64);
            LineCursor lc{d + cuts[pi], d + cuts[pi + 1]};
            std::string_view line, header;
            std::vector<uint64_t> t;
            std::vector<const char*> at;      // where item j's line starts in the block; at[n] = one past the last line
            std::vector<std::pair<size_t, size_t>> seeds;
            bool open = false, plain = true;  // plain: every line of the record is the number's own canonical text + '\n'
                                              // (no sign, no leading zero, no CR): a kept range can be copied as it stands
            auto flush_record = [&]() {
                if (!open) return;
                open = false;
                t.push_back(0);                                                   // :99
                seeds.clear();
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
                if (ranks && !seeds.empty()) {   // :150-163: only the best extended seed; max_by_key keeps the LAST maximum
                    size_t best = 0;
                    uint64_t best_score = 0;
                    for (size_t i = 0; i < seeds.size(); i++) {
                        uint64_t sc = 0;
                        // take(e - s) with s > e wraps in a release build: every taxon from s on, the sentinel included
                        const size_t stop = seeds[i].first > seeds[i].second ? t.size() : std::min(seeds[i].second, t.size());
                        for (size_t j = seeds[i].first; j < stop; j++) sc += score_of(t[j]);
                        if (i == 0 || sc >= best_score) { best = i; best_score = sc; }
                    }
                    seeds[0] = seeds[best];
                    seeds.resize(1);
                }
                const size_t mark = P.out.size();
                P.out.push_back('>'); P.out.append(header); P.out.push_back('\n');
                for (auto& se : seeds) {
                    if (se.first > se.second || se.second > t.size()) { P.panic = true; P.out.resize(mark); break; }   // taxons[start..end]
                    // (the sentinel 0 pushed above has no line: a range that reaches it is formatted)
                    if (se.first == se.second) continue;
                    if (plain && se.second < t.size()) P.out.append(at[se.first], (size_t)(at[se.second] - at[se.first]));
                    else for (size_t j = se.first; j < se.second; j++) { append_uint(P.out, t[j]); P.out.push_back('\n'); }
                }
                t.clear();
                at.clear();
                plain = true;
            };
            try {
                for (;;) {
                    // the common line -- a canonical decimal number directly followed by '\n' -- is read where it stands: the digits
                    // give the value and the line's end in one pass (a memchr per 8-byte line was most of this command's time)
                    while (lc.p < lc.end && !P.panic) {
                        const char* q = lc.p;
                        uint64_t v = 0;
                        unsigned c;
                        while (q < lc.end && (c = (unsigned)(unsigned char)*q - '0') <= 9) { v = v * 10 + c; q++; }
                        const size_t nd = (size_t)(q - lc.p);
                        if (nd < 1 || nd > 18 || q >= lc.end || *q != '\n' || (nd > 1 && *lc.p == '0')) break;   // anything else: the general line below
                        t.push_back(v);
                        if (at.empty()) at.push_back(lc.p);
                        at.push_back(q + 1);
                        lc.p = q + 1;
                    }
                    if (!lc.next(line) || P.panic) break;
                    if (!line_utf8_ok(line)) { flush_record(); P.error = CliError("stream did not contain valid UTF-8"); return; }
                    if (!line.empty() && line[0] == '>') { flush_record(); header = line.substr(1); open = true; continue; }
                    t.push_back(parse_usize(line));
                    // canonical text followed directly by '\n'?  (lc.p is the start of the next line)
                    const char* nxt = lc.p;
                    if (line[0] == '+' || (line[0] == '0' && line.size() > 1) || nxt != line.data() + line.size() + 1 || nxt[-1] != '\n') plain = false;
                    if (at.empty()) at.push_back(line.data());
                    at.push_back(nxt);
                }
                if (!P.panic) flush_record();        // (a panic was noticed at the NEXT record's header, which is open by now: nothing of it is printed)
            } catch (CliError& e) { P.error = e; }   // the failing record is not printed
        });
        for (size_t i = 0; i < np; i++) {
            write_owned_or_copy(stdout, parts[i].out.data(), parts[i].out.size(), latch);
            if (parts[i].panic) { fflush(stdout); throw CliPanic("slice index out of range"); }
            if (parts[i].error) { fflush(stdout); throw *parts[i].error; }
        }
    }
    fflush(stdout);
    return 0;
}

// ---- uniq (src/commands/uniq.rs:56-84) ----------------------------------------------------------
int cmd_uniq(const std::vector<std::string>& args) {
    Parsed p = parse_args(args, {{'s', "separator", true}, {'w', "wrap", false}, {'d', "delimiter", true}});
    std::string sep = p.get("separator").value_or("\n");
    bool wrap = p.has("wrap");
    auto delim = p.get("delimiter");
    if (!p.positional.empty()) usage_error("Found argument '" + p.positional[0] + "' which wasn't expected, or isn't valid in this context");
    if (sep == "\n" && !wrap) {
        // block-parallel tokenization into record descriptors; the merge of consecutive equal headers is a
        // serial pass over descriptors; a record body is copied verbatim (items are already "item\n")
        struct Rec { size_t hdr_off; uint32_t hdr_len; size_t body_off, body_len; uint32_t n_items; bool verbatim; };
        struct alignas(128) UPart { std::vector<Rec> recs; bool bad_utf8 = false; };
        BlockReader blocks(stdin);
        const unsigned T = worker_count();
        const char* d;
        size_t len;
        // two output buffers take turns: a flushed one belongs to the writer thread until it is written (it leaves without a copy)
        std::string out_bufs[2], cur_hdr;  // cur_hdr: the record being merged (survives block boundaries)
        OutLatch out_latches[2];
        int out_turn = 0;
        std::string* out = &out_bufs[0];
        // The body of the open group goes straight to `out` behind its header line -- except while it is still at most ONE item of
        // no bytes, which the writer prints as nothing at all: items.join("\n") is empty iff there is no item or a single empty
        // one (src/io/fasta.rs:168,176-178); that single "\n" waits in `cur_small` until a second item decides.
        uint64_t cur_items = 0, cur_bytes = 0;
        size_t cur_mark = 0;               // where the open group starts in `out`: what lies behind is not final (a read error drops it)
        bool have_cur = false, cur_small = false;
        auto write_cur = [&]() { have_cur = false; cur_small = false; };     // (a group that ends while `cur_small` is pending prints no body)
        auto add_body = [&](const char* b, size_t n, uint64_t items) {
            if (cur_items + items <= 1 && cur_bytes + n <= 1) { cur_small = cur_small || n == 1; }
            else { if (cur_small) { out->push_back('\n'); cur_small = false; } out->append(b, n); }
            cur_items += items;
            cur_bytes += n;
        };
        blocks.map_regular_file();
        std::vector<UPart> parts;
        while (blocks.next(d, len)) {
            if (blocks.first_line_error()) throw CliError("Expected > at beginning of fasta header.");
            std::vector<size_t> cuts = split_records(d, len, T);
            const size_t np = cuts.size() - 1;
            if (parts.size() < np) parts.resize(np);
            parallel_for(np, [&](size_t pi) {
                UPart& P = parts[pi];
                P.recs.clear();
                P.bad_utf8 = false;
                std::vector<Rec> recs = std::move(P.recs);   // (filled on this thread's stack: see LookupPart)
                recs.clear();
                const size_t range_end = cuts[pi + 1];
                bool stop = false;
                scan_lines(d, cuts[pi], range_end, stop, [&](size_t ls, size_t e, bool high) {
                    size_t le = e;
                    const bool cr = le > ls && d[le - 1] == '\r';
                    if (cr) le--;
                    if (high && !valid_utf8(std::string_view(d + ls, le - ls))) { P.bad_utf8 = true; stop = true; return; }
                    const size_t next = e < range_end ? e + 1 : range_end;      // start of the next line
                    if (le > ls && d[ls] == '>') {
                        recs.push_back(Rec{ls + 1, (uint32_t)(le - ls - 1), next, 0, 0, true});
                        return;
                    }
                    Rec& r = recs.back();
                    r.n_items++;
                    r.body_len = next - r.body_off;
                    // verbatim copy needs "item\n" exactly: no stripped '\r', newline-terminated
                    if (cr || e >= range_end) r.verbatim = false;
                });
                P.recs = std::move(recs);
            });
            bool stop = false;
            for (size_t pi = 0; pi < np && !stop; pi++) {
                for (const Rec& r : parts[pi].recs) {
                    std::string_view h(d + r.hdr_off, r.hdr_len);
                    if (delim) { size_t at = h.find(*delim); if (at != std::string_view::npos) h = h.substr(0, at); }
                    if (!(have_cur && h == std::string_view(cur_hdr))) {          // src/commands/uniq.rs:70-77
                        write_cur();
                        cur_hdr.assign(h); cur_items = 0; cur_bytes = 0; have_cur = true;
                        cur_mark = out->size();
                        out->push_back('>'); out->append(cur_hdr); out->push_back('\n');
                    }
                    if (r.verbatim) add_body(d + r.body_off, r.body_len, r.n_items);
                    else {
                        LineCursor lc{d + r.body_off, d + r.body_off + r.body_len};
                        std::string_view line;
                        std::string one;
                        while (lc.next(line)) { one.assign(line); one.push_back('\n'); add_body(one.data(), one.size(), 1); }
                    }
                }
                if (parts[pi].bad_utf8) stop = true;
            }
            {   // the completed groups leave; the open one stays behind (the reference still holds it in `last`: a read error drops it)
                const size_t done = have_cur ? cur_mark : out->size();
                if (done > (1u << 22) || stop) {
                    const int other = out_turn ^ 1;
                    out_latches[other].wait();
                    out_bufs[other].assign(out->data() + done, out->size() - done);      // the open group moves on to the other buffer
                    write_owned_or_copy(stdout, out->data(), done, out_latches[out_turn]);
                    out_turn = other;
                    out = &out_bufs[other];
                    cur_mark = 0;
                }
            }
            if (stop) {
                // the record read so far is NOT written by the reference (it is still held in `last`): only completed ones are
                fflush(stdout);
                throw CliError("stream did not contain valid UTF-8");
            }
        }
        write_cur();                       // EOF completes the open group
        fwrite(out->data(), 1, out->size(), stdout);
        fflush(stdout);
        return 0;
    }
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

// ---- prot2kmer2agg: `prot2kmer2lca -o IX | seedextend -s S -g G | uniq [-d D] | taxa2agg ... TAX` as ONE command ----
// Not a reference subcommand: it is the four-stage 9-mer pipeline of scripts/umgap-analyse.sh:257-310 with the
// three text pipes (84 decimal ids per read pair, parsed and printed three times) removed.  On every input the
// four-stage pipe accepts it prints the same bytes as the pipe (tests/test_cli_gpu.py); the hits stay on the
// device (umgap_classify_records).  Deliberate differences, all on error paths: a failing record (invalid UTF-8,
// a window cutting a character, an unknown taxon) ends the run after the reads that precede it were printed.
// (FusePart and its tokenizer, parse_fuse_part: fuse_parse.hpp)

// one block's worth of reads, self-contained (nothing points into the block buffer)
struct FuseJob {
    // the residues stay where the tokenizer threads wrote them: the carried group's first, then one piece per part of the block
    // (page-locked buffers, umgap_classify_records_sg); `bytes` is only filled with UMGAP_FUSE_PIECES=0 (one array, as before round 5)
    PinnedVec<uint8_t> carry_bytes, bytes;
    std::vector<const uint8_t*> pieces;
    std::vector<uint64_t> piece_ends;
    PinnedVec<uint64_t> frag_off, read_frags;
    std::string hdrs;                 // the reads' headers, back to back
    std::vector<uint64_t> hdr_end;    // n_reads + 1
    size_t n_reads = 0;
    PinnedVec<uint32_t> out_taxon, out_status;
    std::vector<RawBuf> outs;         // formatted output of the job, one buffer per formatting thread
};

int cmd_prot2kmer2agg(const std::vector<std::string>& args) {
    Parsed p = parse_args(args, {{'k', "length", true}, {'o', "one-on-one", false}, {'s', "min-seed-size", true}, {'g', "max-gap-size", true},
                                 {'d', "delimiter", true}, {'m', "method", true}, {'a', "aggregate", true}, {'f', "factor", true},
                                 {'l', "lower-bound", true}, {'r', "ranked", false}, {0, "no-seedextend", false}, {0, "socket", true}});
    if (opt_usize(p, "length", 9) != 9) throw CliError("prot2kmer2agg needs k = 9");
    uint64_t s = opt_usize(p, "min-seed-size", 2), g = opt_usize(p, "max-gap-size", 0);
    bool se = !p.has("no-seedextend");
    // with -s 0 / -s 1 the reference's seedextend can panic on a leading run of misses (slice start > end,
    // src/commands/seedextend.rs:118,145 sliced at :169-173): the device reports such reads with UMGAP_READ_PANIC and the
    // command exits 101 after printing the reads before it, like the reference's pipe does
    if (s > 0xFFFFFFFFull || g > 0xFFFFFFFFull) throw CliError("seed / gap size out of range");
    auto delim = p.get("delimiter");
    std::string m = p.get("method").value_or("tree"), a = p.get("aggregate").value_or("hybrid");
    if (m != "tree" && m != "rmq") usage_error("'" + m + "' isn't a valid value for '--method <method>'\n\t[possible values: rmq, tree]");
    if (a != "lca*" && a != "hybrid" && a != "mrtl") usage_error("'" + a + "' isn't a valid value for '--aggregate <strategy>'\n\t[possible values: hybrid, lca*, mrtl]");
    float factor = opt_f32(p, "factor", 0.25f), lower_bound = opt_f32(p, "lower-bound", 0.0f);
    bool ranked = p.has("ranked");
    if (p.positional.size() != 2) usage_error("prot2kmer2agg takes <fst-file> <taxon-file>");
    int method = m == "tree" ? UMGAP_METHOD_TREE : UMGAP_METHOD_RMQ;
    int strategy = a == "lca*" ? UMGAP_STRATEGY_LCA_STAR : (a == "hybrid" ? UMGAP_STRATEGY_HYBRID : UMGAP_STRATEGY_MRTL);
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const bool timing = getenv("UMGAP_TIMING") != nullptr;   // phase times on stderr
    const double t_begin = now();
    const unsigned T = worker_count();
    // The page-locked buffers of the two stages belong to the command, not to a stream: they are allocated once, BESIDE the index load
    // (page-locking ~100 MB takes 20-30 ms, the load of a real index seconds), serve every connection of the socket server and are
    // freed after the last read is out.  Two sets of parts: the device call of block b reads the residues out of the parts' own
    // buffers while block b + 1 is tokenized.
    FuseJob jobs[2];
    std::vector<FusePart> part_sets[2];
    auto pinned_part = [](FusePart& P) {
        P.bytes.alloc = +[](size_t n) -> void* { void* q = nullptr; return umgap_host_alloc(n, &q) == UMGAP_OK ? q : nullptr; };
        P.bytes.release = +[](void* q) { umgap_host_free(q); };
    };
    std::thread prealloc([&]() {
        if (const char* e = getenv("UMGAP_FUSE_PREALLOC")) if (e[0] == '0') return;      // A/B: buffers allocated by the stream as it goes
        try {
            const size_t block = BlockReader::piped_block(BlockReader::default_block());
            const size_t per_part = block / T + block / (4 * T) + (64u << 10), recs = block / 40 + 4096;   // (a record of 2 x 50 residues: ~66 bytes of text)
            for (auto& parts : part_sets) {
                parts.resize(T);
                for (auto& P : parts) { pinned_part(P); P.bytes.reset_to(per_part); P.cap_bytes = per_part; }
            }
            for (auto& J : jobs) {
                J.frag_off.resize(recs); J.read_frags.resize(recs); J.out_taxon.resize(recs); J.out_status.resize(recs); J.carry_bytes.resize(64u << 10);
            }
        } catch (...) {}      // (no device, no memory: the stream allocates as it goes and reports what it meets)
    });
    struct JoinPrealloc { std::thread& t; ~JoinPrealloc() { if (t.joinable()) t.join(); } } join_prealloc{prealloc};
    umgap_index* ix = open_index(p.positional[0]);
    if (umgap_index_kind(ix) == 2) throw CliError("prot2kmer2agg needs a 9-mer index");
    umgap_taxonomy* tx = load_taxonomy(p.positional[1]);
    if (method == UMGAP_METHOD_RMQ && strategy == UMGAP_STRATEGY_HYBRID)
        fprintf(stderr, "Warning: this is a hybrid between LCA/MRTL, not LCA*/MRTL\n");
    if (method == UMGAP_METHOD_TREE && strategy == UMGAP_STRATEGY_MRTL)
        throw CliError("Invalid invocation: Tree and MaximumRootToLeafPath cannot be combined");
    // index and taxonomy are both here: the table is bound to the taxonomy once (values = preorder interval codes, 0.3 s for a 128 GB
    // table), which is how the fused kernel runs fastest -- the configuration bench.py's headline times.  A table some value of which
    // is no taxon of this taxonomy stays unbound (the unbound kernels report such hits per read, like the reference); UMGAP_BIND=0
    // keeps the taxon ids (A/B).
    {
        const char* env = getenv("UMGAP_BIND");
        if (!(env && env[0] == '0') && umgap_index_bind_taxonomy(ix, tx) != UMGAP_OK && timing)
            fprintf(stderr, "prot2kmer2agg: table not bound to the taxonomy: %s\n", umgap_last_error());
    }
    prealloc.join();
    const double t_start = now();
    auto stream = [&](FILE* in, FILE* outf) {
    double t_read = 0, t_parse = 0, t_group = 0, t_wait = 0, t_dev = 0, t_fmt = 0;

    // stage 2 (its own thread, one job at a time): device call, format, write -- while stage 1 tokenizes the next block
    auto run_job = [&](FuseJob& J) {
        double t0 = now();
        J.out_taxon.resize(J.n_reads);
        J.out_status.resize(J.n_reads);
        if (J.n_reads) LIB(umgap_classify_records_sg(ix, tx, method, strategy, factor, lower_bound, ranked ? 1 : 0, J.pieces.data(), J.piece_ends.data(), J.pieces.size(),
                                                     J.frag_off.data(), J.read_frags[J.n_reads], J.read_frags.data(), J.n_reads, se ? 1 : 0, (uint32_t)s, (uint32_t)g,
                                                     J.out_taxon.data(), J.out_status.data()));
        t_dev += now() - t0; t0 = now();
        size_t bad = J.n_reads;
        for (size_t r = 0; r < J.n_reads; r++)
            if (J.out_status[r] == UMGAP_READ_UNKNOWN_TAXON || J.out_status[r] == UMGAP_READ_PANIC || J.out_status[r] == 4u) { bad = r; break; }
        const size_t nt = std::max<size_t>(1, std::min<size_t>(T / 2, bad / 4096 + 1));
        // every formatting thread writes its range straight into a buffer of its own that is big enough for it (no per-byte capacity
        // checks); the buffers keep their memory from block to block
        std::vector<RawBuf>& outs = J.outs;
        if (outs.size() < nt) outs.resize(nt);
        parallel_for(nt, [&](size_t t) {
            const size_t lo = bad * t / nt, hi = bad * (t + 1) / nt;
            RawBuf& o = outs[t];
            o.need((hi - lo) * 13 + (size_t)(J.hdr_end[hi] - J.hdr_end[lo]) + 16);      // '>' + header + '\n' + at most 10 digits + '\n'
            char* w = o.p.get();
            const char* hdrs = J.hdrs.data();
            for (size_t r = lo; r < hi; r++) {
                const size_t hl = (size_t)(J.hdr_end[r + 1] - J.hdr_end[r]);
                *w++ = '>';
                memcpy(w, hdrs + J.hdr_end[r], hl);
                w += hl;
                *w++ = '\n';
                w = write_uint(w, J.out_taxon[r]);
                *w++ = '\n';
            }
            o.n = (size_t)(w - o.p.get());
        });
        for (size_t t = 0; t < nt; t++) fwrite(outs[t].p.get(), 1, outs[t].n, outf);
        t_fmt += now() - t0;
        if (bad < J.n_reads) {
            fflush(outf);
            if (J.out_status[bad] == UMGAP_READ_UNKNOWN_TAXON) throw CliError("Unknown Taxon ID: " + std::to_string(J.out_taxon[bad]));
            if (J.out_status[bad] == 4u) throw CliError("a read holds more than 1024 distinct taxa: not supported by the device aggregator");
            throw CliPanic("index out of bounds");
        }
    };
    int slot = 0;
    std::thread worker;
    std::exception_ptr worker_error;
    auto join_worker = [&]() {
        if (worker.joinable()) { double t0 = now(); worker.join(); t_wait += now() - t0; }
        if (worker_error) { std::exception_ptr e = worker_error; worker_error = nullptr; std::rethrow_exception(e); }
    };
    struct Joiner { std::thread& t; ~Joiner() { if (t.joinable()) t.join(); } } joiner{worker};   // never leave a running thread behind

    // the group `uniq` is still merging when a block ends (src/commands/uniq.rs:66-79): carried into the next block
    struct Carry { bool have = false; std::string hdr; std::vector<uint8_t> bytes; std::vector<uint64_t> frag_len; } carry;
    const bool gather_first = [] { const char* e = getenv("UMGAP_FUSE_PIECES"); return e && e[0] == '0'; }();
    // 32 MB blocks here also from a mapped file: with two overlapped stages the first block's tokenizing and the last block's device
    // call are not hidden, and 8 M pairs are only 16 blocks of 64 MB (stream 0.217 s against 0.152 s at 32 MB; no difference at 16 M pairs)
    BlockReader blocks(in, BlockReader::piped_block(BlockReader::default_block()));
    blocks.map_regular_file();   // a file on stdin is mapped: blocks are views of the page cache, nothing is copied by this thread
    if (in != stdin) blocks.connection();
    for (;;) {
        const char* d = nullptr;
        size_t len = 0;
        double t0 = now();
        const bool more = blocks.next(d, len);
        t_read += now() - t0; t0 = now();
        if (more && blocks.first_line_error()) { join_worker(); throw CliError("Expected > at beginning of fasta header."); }
        std::vector<size_t> cuts = more ? split_records(d, len, T) : std::vector<size_t>{0};
        const size_t np = cuts.size() - 1;
        std::vector<FusePart>& parts = part_sets[slot];     // (job `slot` of two blocks ago is through: join_worker below, before the last launch)
        if (parts.size() < np) {                   // the parts keep their buffers from block to block (fresh pages cost more than the parse)
            const size_t had = parts.size();
            parts.resize(np);
            for (size_t i = had; i < np; i++) pinned_part(parts[i]);
        }
        parallel_for(np, [&](size_t i) { parse_fuse_part(d, cuts[i], cuts[i + 1], delim, parts[i]); });
        t_parse += now() - t0; t0 = now();
        size_t used = np;
        for (size_t i = 0; i < np; i++)
            if (parts[i].bad_utf8 || parts[i].panic) { used = i + 1; break; }
        const bool failed = used > 0 && (parts[used - 1].bad_utf8 || parts[used - 1].panic);
        // layout: the carried group first, then the block's records in input order; groups across range seams are merged
        FuseJob& J = jobs[slot];
        size_t n_frag = carry.frag_len.size(), n_bytes = carry.bytes.size(), n_groups = carry.have ? 1 : 0, n_hdr = 0;
        std::string_view prev = carry.hdr;
        bool have_prev = carry.have;
        for (size_t i = 0; i < used; i++) {
            FusePart& P = parts[i];
            P.rec_base = n_frag; P.byte_base = n_bytes; P.group_base = n_groups; P.hdr_base = n_hdr;
            n_frag += P.frag_len.size(); n_bytes += P.n_bytes;
            if (P.hdr_off.empty()) continue;
            P.merged = have_prev && P.header(d, 0) == prev;
            n_groups += P.gstart.size() - (P.merged ? 1 : 0);
            n_hdr += P.hdr_bytes - (P.merged ? P.hdr_len[0] : 0);
            prev = P.header(d, P.hdr_off.size() - 1);
            have_prev = true;
        }
        J.pieces.clear();
        J.piece_ends.clear();
        if (!carry.bytes.empty()) {
            J.carry_bytes.resize(carry.bytes.size());
            memcpy(J.carry_bytes.data(), carry.bytes.data(), carry.bytes.size());
            J.pieces.push_back(J.carry_bytes.data());
            J.piece_ends.push_back(carry.bytes.size());
        }
        for (size_t i = 0; i < used; i++)
            if (parts[i].n_bytes) { J.pieces.push_back(parts[i].bytes.get()); J.piece_ends.push_back(parts[i].byte_base + parts[i].n_bytes); }
        // bytes [a, a + len) of the concatenation
        auto copy_bytes = [&](uint8_t* dst, uint64_t a, uint64_t len) {
            size_t i = (size_t)(std::upper_bound(J.piece_ends.begin(), J.piece_ends.end(), a) - J.piece_ends.begin());
            for (uint64_t at = 0; at < len && i < J.pieces.size(); i++) {
                const uint64_t p0 = i ? J.piece_ends[i - 1] : 0, take = std::min(len - at, J.piece_ends[i] - (a + at));
                memcpy(dst + at, J.pieces[i] + (a + at - p0), take);
                at += take;
            }
        };
        J.frag_off.resize(n_frag + 1);
        J.read_frags.resize(n_groups + 1);
        J.hdr_end.resize(n_groups + 1);
        const size_t carry_hdr = carry.have ? carry.hdr.size() : 0;
        J.hdrs.resize(carry_hdr + n_hdr);
        J.frag_off[0] = 0;
        for (size_t f = 0; f < carry.frag_len.size(); f++) J.frag_off[f + 1] = J.frag_off[f] + carry.frag_len[f];
        if (carry.have) { J.read_frags[0] = 0; J.hdr_end[0] = 0; memcpy(J.hdrs.data(), carry.hdr.data(), carry_hdr); }
        parallel_for(used, [&](size_t i) {
            const FusePart& P = parts[i];
            uint64_t o = P.byte_base;
            for (size_t r = 0; r < P.frag_len.size(); r++) { o += P.frag_len[r]; J.frag_off[P.rec_base + r + 1] = o; }
            size_t gi = P.group_base, ho = carry_hdr + P.hdr_base;
            for (size_t k = P.merged ? 1 : 0; k < P.gstart.size(); k++, gi++) {
                size_t r = P.gstart[k];
                J.read_frags[gi] = P.rec_base + r;
                J.hdr_end[gi] = ho;
                memcpy(J.hdrs.data() + ho, d + P.hdr_off[r], P.hdr_len[r]);
                ho += P.hdr_len[r];
            }
        });
        J.read_frags[n_groups] = n_frag;
        J.hdr_end[n_groups] = J.hdrs.size();
        size_t n_now = n_groups;
        Carry next;
        if (more && !failed && n_groups > 0) {   // the last group may continue in the next block
            n_now = n_groups - 1;
            next.have = true;
            next.hdr.assign(J.hdrs.data() + J.hdr_end[n_now], J.hdr_end[n_groups] - J.hdr_end[n_now]);
            size_t f0 = J.read_frags[n_now];
            next.bytes.resize(J.frag_off[n_frag] - J.frag_off[f0]);
            copy_bytes(next.bytes.data(), J.frag_off[f0], next.bytes.size());
            for (size_t f = f0; f < n_frag; f++) next.frag_len.push_back(J.frag_off[f + 1] - J.frag_off[f]);
        }
        J.n_reads = n_now;
        if (gather_first) {          // A/B: the pieces gathered into one page-locked array by this stage, as before round 5
            J.bytes.resize(n_bytes + 1);
            size_t k = 0;
            if (!carry.bytes.empty()) k = 1;
            parallel_for(used, [&](size_t i) { if (parts[i].n_bytes) memcpy(J.bytes.data() + parts[i].byte_base, parts[i].bytes.get(), parts[i].n_bytes); });
            if (k) memcpy(J.bytes.data(), J.carry_bytes.data(), carry.bytes.size());
            J.pieces.assign(1, J.bytes.data());
            J.piece_ends.assign(1, n_bytes);
        }
        // taxa2agg's reader strips one more CR from the header line it prints (looked for by all threads: 16 M reads per second and
        // thread otherwise stood in front of every block)
        std::atomic<bool> any_cr{false};
        {
            const size_t nt = std::max<size_t>(1, std::min<size_t>(T, n_now / 65536 + 1));
            parallel_for(nt, [&](size_t t) {
                bool cr = false;
                for (size_t r = n_now * t / nt, hi = n_now * (t + 1) / nt; r < hi && !cr; r++)
                    cr = J.hdr_end[r + 1] > J.hdr_end[r] && J.hdrs[J.hdr_end[r + 1] - 1] == '\r';
                if (cr) any_cr.store(true, std::memory_order_relaxed);
            });
        }
        for (size_t r = 0; any_cr.load() && r < n_now; r++)
            if (J.hdr_end[r + 1] > J.hdr_end[r] && J.hdrs[J.hdr_end[r + 1] - 1] == '\r') {
                // rare: rebuild the header table without those bytes
                std::string h2; std::vector<uint64_t> e2(n_groups + 1, 0);
                for (size_t q = 0; q < n_groups; q++) {
                    std::string_view h(J.hdrs.data() + J.hdr_end[q], J.hdr_end[q + 1] - J.hdr_end[q]);
                    if (q < n_now && !h.empty() && h.back() == '\r') h.remove_suffix(1);
                    e2[q] = h2.size(); h2.append(h);
                }
                e2[n_groups] = h2.size();
                J.hdrs.swap(h2); J.hdr_end.swap(e2);
                break;
            }
        t_group += now() - t0;
        join_worker();                                   // the previous block is out (or its error is raised here)
        carry = std::move(next);
        worker = std::thread([&, slot]() { try { run_job(jobs[slot]); } catch (...) { worker_error = std::current_exception(); } });
        slot ^= 1;
        if (failed) {
            join_worker();
            fflush(outf);
            if (parts[used - 1].panic) throw CliPanic("byte index is not a char boundary");
            throw CliError("stream did not contain valid UTF-8");
        }
        if (!more) break;
    }
    join_worker();
    fflush(outf);
    if (timing) fprintf(stderr, "prot2kmer2agg: open %.3f s | stage 1: read %.3f, tokenize %.3f, group %.3f, wait for stage 2 %.3f | stage 2: device %.3f, format+write %.3f\n",
                        t_start - t_begin, t_read, t_parse, t_group, t_wait, t_dev, t_fmt);
    };
    if (auto sock = p.get("socket")) {
        // the index and the taxonomy stay resident; connections are served one after another, with the status
        // lines of `prot2kmer2lca -s` (src/commands/prot2kmer2lca.rs:116-137) on stdout
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
            const bool async_io = umgap_host::AsyncIo::enabled();
            FILE* in = async_io ? umgap_host::AsyncIo::reader(c, true) : fdopen(c, "rb");
            FILE* outf = async_io ? umgap_host::AsyncIo::writer(dup(c), true) : fdopen(dup(c), "wb");
            try {
                stream(in, outf);
                printf("Connection finished succesfully.\n");
            } catch (CliError& e) { printf("Connection died with an error: %s\n", e.what()); }
            catch (CliPanic& e) { printf("Connection died with an error: %s\n", e.what()); }
            fflush(stdout);
            fclose(outf);
            fclose(in);
        }
    }
    stream(stdin, stdout);
    const double t_streamed = now();
    fflush(stdout);
    umgap_host::AsyncIo::drain_all();
    const double t_drained = now();
    umgap_taxonomy_free(tx);
    umgap_index_free(ix);
    if (timing) fprintf(stderr, "prot2kmer2agg: stream %.3f s, output drained after %.3f s more, handles freed after %.3f s more\n", t_streamed - t_start, t_drained - t_streamed,
                        now() - t_drained);
    return 0;
}

// ---- buildindex: TSV (key<TAB>taxon, sorted) on stdin -> UMGAPIX1 table on stdout (src/commands/buildindex.rs:32-48) ----
// Two passes over the input, both block-parallel, host memory O(block) for a 9-mer index of any size:
//   pass 1 counts the keys and checks what fst::MapBuilder::insert checks (strictly increasing keys) plus the field
//          syntax; stdin is re-read if it is a seekable file, otherwise it is spooled to a temporary file on the way;
//   pass 2 packs the 9-mers and streams them into the device table (umgap_index_builder_*).
// If some key is not [A-Z]{9} the variable-length table is built instead (in memory: tryptic indexes are small).
struct alignas(128) BuildPart {
    size_t n = 0;
    bool all9 = true;
    std::string first, last;                 // first / last key of the range (order check across ranges)
    std::optional<CliError> error;           // first error of the range
    std::vector<uint64_t> keys;
    std::vector<uint32_t> vals;
    StringBatch strs;
};

void parse_index_part(const char* d, size_t begin, size_t end, int pass, bool want_strings, BuildPart& P) {
    LineCursor lc{d + begin, d + end};
    std::string_view l, prev;
    while (lc.next(l)) {
        if (l.empty()) continue;
        size_t tab = l.find('\t');
        if (tab == std::string_view::npos || l.find('\t', tab + 1) != std::string_view::npos) { P.error = CliError("CSV deserialize error: expected two fields"); return; }
        std::string_view k = l.substr(0, tab);
        uint64_t v;
        try { v = parse_usize(l.substr(tab + 1)); } catch (CliError&) { P.error = CliError("CSV deserialize error: invalid integer"); return; }
        if (pass == 1) {
            // fst::MapBuilder::insert wants strictly increasing keys (src/commands/buildindex.rs:42)
            if (P.n && !(prev < k)) { P.error = CliError(std::string(prev == k ? "Error inserting duplicate key: " : "Error inserting out-of-order key: ") + std::string(k)); return; }
            if (k.empty() || k.size() > 63) { P.error = CliError("key '" + std::string(k) + "': keys must be 1..63 bytes long"); return; }
            if (v > 0xFFFFFFFFull) { P.error = CliError("taxon id does not fit 32 bits"); return; }
            if (P.n == 0) P.first.assign(k);
            prev = k;
        }
        uint64_t key = 0;
        const bool nine = k.size() == 9 && umgap_pack9(k.data(), &key);
        if (!nine) P.all9 = false;
        if (pass == 2) {
            if (want_strings) { P.strs.add(k); P.vals.push_back((uint32_t)v); }
            else { P.keys.push_back(key); P.vals.push_back((uint32_t)v); }
        }
        P.n++;
    }
    if (pass == 1 && P.n) P.last.assign(prev);
}

int cmd_buildindex(const std::vector<std::string>& args) {
    Parsed p = parse_args(args, {{'l', "load-factor", true}, {'p', "proteome", true}});
    float load = opt_f32(p, "load-factor", 0.0f);
    const unsigned T = worker_count();
    // -p / --proteome FILE (not in the reference): a protein FASTA; the 9-mer index is then laid out in WALK order along it
    // (include/umgap_hip.h: umgap_walk_builder_*) -- same keys, same values, same answers from every command
    const bool walk = p.has("proteome");
    FILE* proteome = nullptr;
    if (walk) {
        const std::string path = p.get("proteome").value_or("");
        proteome = fopen(path.c_str(), "rb");
        if (!proteome) throw CliError("cannot open " + path + ": " + strerror(errno));
    }
    // --proteome, round 5: no values table on the device.  The proteome first -- every 9-mer of it gets an entry of the walk array --
    // then the TSV, streamed ONCE, values its keys (umgap_walk_builder_new_deferred: position table + walk array resident, nothing
    // else: the real 8 G-key index with the reference's own values fits one GPU).  UMGAP_WALK_VALUES_FIRST=1 keeps the first form
    // (values table built first; keys must all be 9-mers either way).
    if (walk && !getenv("UMGAP_WALK_VALUES_FIRST")) {
        umgap_walk_builder* wb = nullptr;
        umgap_index* wx = nullptr;
        try {
            // an upper bound on the distinct 9-mers of the proteome: its windows (one pass over the file, no device work)
            uint64_t windows = 0;
            {
                FastaReader fr(proteome, true);
                Record rec;
                while (fr.read_record(rec)) { const size_t n = rec.sequence[0].size(); if (n >= 9) windows += n - 8; }
                rewind(proteome);
            }
            LIB(umgap_walk_builder_new_deferred(std::max<uint64_t>(windows, 1), load, 0, &wb));
            FastaReader fr(proteome, true);
            Record rec;
            std::vector<uint8_t> bytes;
            std::vector<uint64_t> offs{0};
            uint32_t max_len = 9;
            auto flush = [&]() {
                if (offs.size() > 1) LIB(umgap_walk_builder_add(wb, bytes.data(), offs.data(), offs.size() - 1, max_len));
                bytes.clear();
                offs.assign(1, 0);
                max_len = 9;
            };
            while (fr.read_record(rec)) {
                const std::string& sq = rec.sequence[0];
                if (sq.size() >= 0xFFFFFF00ull) throw CliError("a protein of 4 GiB or more");
                if (bytes.size() + sq.size() > (256u << 20)) flush();
                bytes.insert(bytes.end(), sq.begin(), sq.end());
                offs.push_back(bytes.size());
                max_len = std::max<uint32_t>(max_len, (uint32_t)sq.size());
            }
            flush();
            // the TSV: what fst::MapBuilder::insert checks (strictly increasing keys) + the field syntax, block-parallel, one pass
            BlockReader blocks(stdin);
            blocks.lines_only();
            const char* d;
            size_t len;
            std::string last_key;
            bool have_last = false;
            while (blocks.next(d, len)) {
                std::vector<size_t> cuts = split_lines(d, len, T);
                const size_t np = cuts.size() - 1;
                std::vector<BuildPart> parts(np);
                parallel_for(np, [&](size_t i) {
                    parse_index_part(d, cuts[i], cuts[i + 1], 1, false, parts[i]);
                    if (!parts[i].error && parts[i].all9) { BuildPart P2; parse_index_part(d, cuts[i], cuts[i + 1], 2, false, P2); parts[i].keys.swap(P2.keys); parts[i].vals.swap(P2.vals); }
                });
                for (size_t i = 0; i < np; i++) {
                    BuildPart& P = parts[i];
                    if (P.n && have_last && !(std::string_view(last_key) < std::string_view(P.first)))
                        throw CliError(std::string(last_key == P.first ? "Error inserting duplicate key: " : "Error inserting out-of-order key: ") + P.first);
                    if (P.error) throw *P.error;
                    if (!P.all9) throw CliError("--proteome needs an index of 9-mers");
                    if (P.n) { last_key = P.last; have_last = true; }
                    for (uint32_t v : P.vals) if (v == 0) throw CliError("--proteome: a value of 0 cannot be told from \"no key\" in a walk-order index");
                    LIB(umgap_walk_builder_set_values(wb, P.keys.data(), P.vals.data(), P.keys.size()));
                }
            }
            LIB(umgap_walk_builder_finish(wb, &wx));
            wb = nullptr;
        } catch (...) {
            if (wb) umgap_walk_builder_free(wb);
            fclose(proteome);
            throw;
        }
        fclose(proteome);
        LIB(umgap_index_save(wx, "/dev/stdout"));
        umgap_index_free(wx);
        return 0;
    }
    // pass 1 (spooling unless stdin can be rewound)
    const bool seekable = lseek(fileno(stdin), 0, SEEK_CUR) != (off_t)-1;
    FILE* spool = nullptr;
    if (!seekable) {
        spool = tmpfile();
        if (!spool) throw CliError(std::string("cannot create a temporary file for the second pass: ") + strerror(errno));
    }
    size_t n_keys = 0;
    bool all9 = true;
    std::string last_key;
    bool have_last = false;
    {
        BlockReader blocks(stdin);
        blocks.lines_only();
        const char* d;
        size_t len;
        while (blocks.next(d, len)) {
            if (spool && fwrite(d, 1, len, spool) != len) throw CliError(std::string("writing the temporary file: ") + strerror(errno));
            std::vector<size_t> cuts = split_lines(d, len, T);
            const size_t np = cuts.size() - 1;
            std::vector<BuildPart> parts(np);
            parallel_for(np, [&](size_t i) { parse_index_part(d, cuts[i], cuts[i + 1], 1, false, parts[i]); });
            for (size_t i = 0; i < np; i++) {
                BuildPart& P = parts[i];
                if (P.n && have_last && !(std::string_view(last_key) < std::string_view(P.first)))
                    throw CliError(std::string(last_key == P.first ? "Error inserting duplicate key: " : "Error inserting out-of-order key: ") + P.first);
                if (P.error) throw *P.error;
                if (P.n) { last_key = P.last; have_last = true; }
                n_keys += P.n;
                all9 = all9 && P.all9;
            }
        }
    }
    FILE* in2 = stdin;
    if (spool) { fflush(spool); rewind(spool); in2 = spool; }
    else if (fseeko(stdin, 0, SEEK_SET) != 0) throw CliError("cannot rewind the input for the second pass");
    else clearerr(stdin);
    // pass 2
    umgap_index* ix = nullptr;
    BlockReader blocks(in2);
    blocks.lines_only();
    const char* d;
    size_t len;
    if (all9) {
        umgap_index_builder* b = nullptr;
        LIB(umgap_index_builder_new(n_keys, load, 0, &b));
        try {
            while (blocks.next(d, len)) {
                std::vector<size_t> cuts = split_lines(d, len, T);
                const size_t np = cuts.size() - 1;
                std::vector<BuildPart> parts(np);
                parallel_for(np, [&](size_t i) { parse_index_part(d, cuts[i], cuts[i + 1], 2, false, parts[i]); });
                for (size_t i = 0; i < np; i++) LIB(umgap_index_builder_add(b, parts[i].keys.data(), parts[i].vals.data(), parts[i].keys.size()));
            }
            LIB(umgap_index_builder_finish(b, &ix));
        } catch (...) { umgap_index_builder_free(b); throw; }
    } else {
        StringBatch strs;
        std::vector<uint32_t> vals;
        while (blocks.next(d, len)) {
            BuildPart P;
            parse_index_part(d, 0, len, 2, true, P);
            for (size_t i = 0; i < P.strs.size(); i++) strs.add(std::string_view((const char*)P.strs.bytes.data() + P.strs.offsets[i], P.strs.offsets[i + 1] - P.strs.offsets[i]));
            vals.insert(vals.end(), P.vals.begin(), P.vals.end());
        }
        LIB(umgap_index_build_var(strs.bytes.data(), strs.offsets.data(), vals.data(), vals.size(), load, 0, &ix));
    }
    if (spool) fclose(spool);
    if (walk) {
        if (umgap_index_kind(ix) != 9) { umgap_index_free(ix); throw CliError("--proteome needs an index of 9-mers"); }
        umgap_walk_builder* wb = nullptr;
        umgap_index* wx = nullptr;
        try {
            LIB(umgap_walk_builder_new(ix, load, &wb));
            FastaReader fr(proteome, true);
            Record rec;
            std::vector<uint8_t> bytes;
            std::vector<uint64_t> offs{0};
            uint32_t max_len = 9;
            auto flush = [&]() {
                if (offs.size() > 1) LIB(umgap_walk_builder_add(wb, bytes.data(), offs.data(), offs.size() - 1, max_len));
                bytes.clear();
                offs.assign(1, 0);
                max_len = 9;
            };
            while (fr.read_record(rec)) {
                const std::string& sq = rec.sequence[0];
                if (sq.size() >= 0xFFFFFF00ull) throw CliError("a protein of 4 GiB or more");
                if (bytes.size() + sq.size() > (256u << 20)) flush();
                bytes.insert(bytes.end(), sq.begin(), sq.end());
                offs.push_back(bytes.size());
                max_len = std::max<uint32_t>(max_len, (uint32_t)sq.size());
            }
            flush();
            LIB(umgap_walk_builder_finish(wb, &wx));
            wb = nullptr;
        } catch (...) {
            if (wb) umgap_walk_builder_free(wb);
            umgap_index_free(ix);
            fclose(proteome);
            throw;
        }
        fclose(proteome);
        umgap_index_free(ix);
        ix = wx;
    }
    LIB(umgap_index_save(ix, "/dev/stdout"));
    umgap_index_free(ix);
    return 0;
}

// ---- printindex: the key/value pairs of an index as TSV, sorted by key (src/commands/printindex.rs:38-51) ----
int cmd_printindex(const std::vector<std::string>& args) {
    Parsed p = parse_args(args, {});
    std::string path = need_positional(p, "fst-file");
    umgap_index* ix = open_index(path);
    Out out(stdout);
    size_t n = 0;
    if (umgap_index_kind(ix) != 2) {
        size_t cap = (size_t)umgap_index_num_keys(ix);
        std::vector<uint64_t> keys(cap);
        std::vector<uint32_t> taxa(cap);
        LIB(umgap_index_dump9(ix, keys.data(), taxa.data(), cap, &n));
        if (n > cap) throw CliError("index holds more keys than its header says");
        std::vector<size_t> order(n);
        for (size_t i = 0; i < n; i++) order[i] = i;
        std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return keys[a] < keys[b]; });  // packed order = byte order of [A-Z]{9}
        char nine[9];
        for (size_t i : order) {
            umgap_unpack9(keys[i], nine);
            out.buf.append(nine, 9); out.buf.push_back('\t'); append_u64(out.buf, taxa[i]); out.buf.push_back('\n');
            out.maybe_flush();
        }
    } else {
        size_t cap = (size_t)umgap_index_num_keys(ix);
        uint64_t heap_bytes = 0;
        std::vector<uint64_t> offs(cap);
        std::vector<uint32_t> lens(cap), taxa(cap);
        LIB(umgap_index_dump_var(ix, nullptr, 0, offs.data(), lens.data(), taxa.data(), cap, &n, &heap_bytes));
        std::vector<uint8_t> heap(heap_bytes ? heap_bytes : 1);
        LIB(umgap_index_dump_var(ix, heap.data(), heap.size(), offs.data(), lens.data(), taxa.data(), cap, &n, &heap_bytes));
        if (n > cap) throw CliError("index holds more keys than its header says");
        std::vector<size_t> order(n);
        for (size_t i = 0; i < n; i++) order[i] = i;
        auto key = [&](size_t i) { return std::string_view((const char*)heap.data() + offs[i], lens[i]); };
        std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return key(a) < key(b); });
        for (size_t i : order) {
            out.buf.append(key(i)); out.buf.push_back('\t'); append_u64(out.buf, taxa[i]); out.buf.push_back('\n');
            out.maybe_flush();
        }
    }
    out.flush();
    umgap_index_free(ix);
    return 0;
}

}  // namespace

static int run_command(std::vector<std::string> args);

int main(int argc, char** argv) {
    std::vector<std::string> args(argv + 1, argv + argc);
    // stdin / stdout behind a reader and a writer thread (async_io.hpp): the stage's pace in a pipe chain is then the slowest of
    // read, work and write instead of their sum.  Everything written is out before the process ends, whichever way it ends.
    const bool worker = !args.empty() && args[0] == "__shard-worker";   // (no data on its stdin / stdout)
    if (!worker && umgap_host::AsyncIo::enabled()) {
        FILE* in = umgap_host::AsyncIo::reader(0);
        FILE* out = umgap_host::AsyncIo::writer(1);
        if (in && out) {
            stdin = in;
            stdout = out;
            atexit([]() { fflush(stdout); umgap_host::AsyncIo::drain_all(); });
        }
    }
    signal(SIGPIPE, SIG_IGN);          // a closed pipe is a failed write (EPIPE) and an `Error: Broken pipe (os error 32)`, as in the reference, not a signal
    int rc = run_command(std::move(args));
    fflush(stdout);
    umgap_host::AsyncIo::drain_all();
    if (rc == 0) {
        int werr = umgap_host::AsyncIo::first_write_error();
        if (!werr && ferror(stdout)) werr = errno ? errno : EIO;
        if (werr) { fprintf(stderr, "Error: %s (os error %d)\n", strerror(werr), werr); rc = 1; }
    }
    return rc;
}

static int run_command(std::vector<std::string> args) {
    try {
        if (args.empty()) usage_error("umgap: a subcommand is required (pept2lca, prot2kmer2lca, prot2tryp2lca, taxa2agg, seedextend, uniq, prot2kmer2agg, buildindex, printindex)");
        std::string cmd = args[0];
        args.erase(args.begin());
        if (cmd == "-V" || cmd == "--version") { printf("umgap %s\n", umgap_version()); return 0; }  // scripts/umgap-analyse.sh:107
        if (cmd == "pept2lca") return cmd_pept2lca(args);
        if (cmd == "prot2kmer2lca") return cmd_prot2kmer2lca(args);
        if (cmd == "prot2tryp2lca") return cmd_prot2tryp2lca(args);
        if (cmd == "taxa2agg") return cmd_taxa2agg(args);
        if (cmd == "prot2kmer2agg") return cmd_prot2kmer2agg(args);
        if (cmd == "seedextend") return cmd_seedextend(args);
        if (cmd == "uniq") return cmd_uniq(args);
        if (cmd == "buildindex") return cmd_buildindex(args);
        if (cmd == "printindex") return cmd_printindex(args);
        if (cmd == "__shard-worker") return cmd_shard_worker(args);
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
