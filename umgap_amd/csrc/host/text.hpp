// text.hpp — host-side text layer of the `umgap` command: FASTA stream reader/writer, the
// taxonomy TSV parser and the number parsing rules the reference inherits from Rust's std.
// Mirrors (behaviour, messages and quirks; not code) src/io/fasta.rs, src/taxon.rs:65-128,
// src/rank.rs:9-42 of unipept/umgap v1.1.1.  Pure C++17, no GPU.
#pragma once
#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <optional>
#include <stdexcept>
#include <string>
#include <string_view>
#include <vector>

namespace umgap_host {

// `Error: <msg>` on stderr, exit 1 (error_chain's quick_main!, src/main.rs:8)
struct CliError : std::runtime_error {
    std::vector<std::string> causes;
    explicit CliError(const std::string& m) : std::runtime_error(m) {}
};
// Rust panic: exit 101
struct CliPanic : std::runtime_error {
    explicit CliPanic(const std::string& m) : std::runtime_error(m) {}
};

// ---- str::parse::<usize>() ---------------------------------------------------------------
inline uint64_t parse_usize(std::string_view s) {
    if (s.empty()) throw CliError("cannot parse integer from empty string");
    size_t i = 0;
    if (s[0] == '+' || s[0] == '-') {
        if (s.size() == 1 || s[0] == '-') throw CliError("invalid digit found in string");
        i = 1;
    }
    uint64_t v = 0;
    if (s.size() - i <= 19) {          // cannot overflow 64 bits: no check per digit (taxon ids have 1-7 digits)
        for (; i < s.size(); i++) {
            unsigned c = (unsigned char)s[i] - '0';
            if (c > 9) throw CliError("invalid digit found in string");
            v = v * 10 + c;
        }
        return v;
    }
    for (; i < s.size(); i++) {
        unsigned c = (unsigned char)s[i] - '0';
        if (c > 9) throw CliError("invalid digit found in string");
        if (v > (UINT64_MAX - c) / 10) throw CliError("number too large to fit in target type");
        v = v * 10 + c;
    }
    return v;
}

// ---- str::parse::<f32>() -----------------------------------------------------------------
inline float parse_f32(std::string_view s) {
    if (s.empty()) throw CliError("cannot parse float from empty string");
    std::string str(s);
    size_t i = (str[0] == '+' || str[0] == '-') ? 1 : 0;
    std::string rest = str.substr(i), low = rest;
    for (auto& c : low) c = (char)tolower((unsigned char)c);
    if (low == "inf" || low == "infinity" || low == "nan") {
        float v = low == "nan" ? __builtin_nanf("") : __builtin_inff();
        return str[0] == '-' ? -v : v;
    }
    size_t j = 0, nd = 0;
    auto digit = [&](size_t k) { return k < rest.size() && rest[k] >= '0' && rest[k] <= '9'; };
    while (digit(j)) { j++; nd++; }
    if (j < rest.size() && rest[j] == '.') { j++; while (digit(j)) { j++; nd++; } }
    bool ok = nd > 0;
    if (ok && j < rest.size() && (rest[j] == 'e' || rest[j] == 'E')) {
        j++;
        if (j < rest.size() && (rest[j] == '+' || rest[j] == '-')) j++;
        size_t ne = 0;
        while (digit(j)) { j++; ne++; }
        ok = ne > 0;
    }
    if (!ok || j != rest.size()) throw CliError("invalid float literal");
    return strtof(str.c_str(), nullptr);
}

inline bool valid_utf8(std::string_view s) {
    const unsigned char* p = (const unsigned char*)s.data();
    size_t n = s.size(), i = 0;
    while (i < n) {
        unsigned char c = p[i];
        if (c < 0x80) { i++; continue; }
        size_t len; uint32_t cp, mn;
        if ((c & 0xE0) == 0xC0) { len = 2; cp = c & 0x1F; mn = 0x80; }
        else if ((c & 0xF0) == 0xE0) { len = 3; cp = c & 0x0F; mn = 0x800; }
        else if ((c & 0xF8) == 0xF0) { len = 4; cp = c & 0x07; mn = 0x10000; }
        else return false;
        if (i + len > n) return false;
        for (size_t k = 1; k < len; k++) {
            if ((p[i + k] & 0xC0) != 0x80) return false;
            cp = (cp << 6) | (p[i + k] & 0x3F);
        }
        if (cp < mn || cp > 0x10FFFF || (cp >= 0xD800 && cp <= 0xDFFF)) return false;
        i += len;
    }
    return true;
}

// ---- line source: BufRead::lines() semantics over a FILE* with a large buffer (src/io/fasta.rs:14) ----
class LineSource {
  public:
    explicit LineSource(FILE* f) : f_(f), buf_(10'000'000) {}
    // false at EOF; the view stays valid until the next call.  '\n' and one trailing '\r' stripped.
    bool next(std::string_view& line) {
        for (;;) {
            const char* base = buf_.data();
            const char* nl = (const char*)memchr(base + pos_, '\n', end_ - pos_);
            if (nl) {
                size_t s = pos_, e = (size_t)(nl - base);
                pos_ = e + 1;
                if (e > s && base[e - 1] == '\r') e--;
                line = std::string_view(base + s, e - s);
                check_utf8(line);
                return true;
            }
            if (eof_) {
                if (pos_ == end_) return false;
                size_t s = pos_, e = end_;
                pos_ = end_;
                if (e > s && base[e - 1] == '\r') e--;
                line = std::string_view(base + s, e - s);
                check_utf8(line);
                return true;
            }
            // refill: move the partial line to the front, grow if it fills the buffer
            if (pos_ > 0) { memmove(buf_.data(), buf_.data() + pos_, end_ - pos_); end_ -= pos_; pos_ = 0; }
            if (end_ == buf_.size()) buf_.resize(buf_.size() * 2);
            size_t got = fread(buf_.data() + end_, 1, buf_.size() - end_, f_);
            if (got == 0) {
                if (ferror(f_)) throw CliError(std::string(strerror(errno)));
                eof_ = true;
            }
            end_ += got;
        }
    }

  private:
    static void check_utf8(std::string_view l) {
        for (unsigned char c : l)
            if (c >= 0x80) {
                if (!valid_utf8(l)) throw CliError("stream did not contain valid UTF-8");
                return;
            }
    }
    FILE* f_;
    std::vector<char> buf_;
    size_t pos_ = 0, end_ = 0;
    bool eof_ = false;
};

// ---- FASTA records (src/io/fasta.rs:38-67): header without '>', items = the following lines ----
struct Record {
    std::string header;
    std::vector<std::string> sequence;
};

class FastaReader {
  public:
    FastaReader(FILE* f, bool unwrap) : lines_(f), unwrap_(unwrap) {}
    // Ok(None) -> false.  Errors throw (the failing line is consumed, like the reference).
    bool read_record(Record& rec) {
        std::string_view l;
        if (deferred_error_) throw CliError("stream did not contain valid UTF-8");
        if (have_) { have_ = false; l = pending_; }
        else if (!lines_.next(l)) return false;
        if (l.empty() || l[0] != '>') throw CliError("Expected > at beginning of fasta header.");
        rec.header.assign(l.substr(1));
        rec.sequence.clear();
        std::string cat;
        for (;;) {
            bool more;
            try { more = lines_.next(l); }
            catch (CliError&) { deferred_error_ = true; break; }  // surfaces at the NEXT read_record (peek semantics)
            if (!more) break;
            if (!l.empty() && l[0] == '>') { pending_.assign(l); have_ = true; break; }
            if (unwrap_) cat.append(l); else rec.sequence.emplace_back(l);
        }
        if (unwrap_) rec.sequence.push_back(std::move(cat));
        return true;
    }
    bool has_deferred_error() const { return deferred_error_; }

  private:
    LineSource lines_;
    bool unwrap_;
    bool have_ = false, deferred_error_ = false;
    std::string pending_;
};

// fasta::Writer (src/io/fasta.rs:141-181) into a growing byte buffer
inline void write_record(std::string& out, std::string_view header, const std::vector<std::string>& items,
                         std::string_view separator, bool wrap) {
    out.push_back('>');
    out.append(header);
    std::string seq;
    for (size_t i = 0; i < items.size(); i++) {
        if (i) seq.append(separator);
        seq.append(items[i]);
    }
    if (!wrap) {
        out.push_back('\n');
        out.append(seq);
    } else {
        for (size_t i = 0; i < seq.size(); i += 70) { out.push_back('\n'); out.append(seq, i, 70); }
    }
    if (!seq.empty()) out.push_back('\n');
}

// ---- taxonomy TSV (src/taxon.rs:89-128) ---------------------------------------------------
static const char* const RANK_NAMES[32] = {
    "no rank", "superkingdom", "domain", "realm", "kingdom", "subkingdom", "superphylum", "phylum",
    "subphylum", "superclass", "class", "subclass", "infraclass", "superorder", "order", "suborder",
    "infraorder", "parvorder", "superfamily", "family", "subfamily", "tribe", "subtribe", "genus",
    "subgenus", "species group", "species subgroup", "species", "subspecies", "varietas", "forma", "strain"};

struct TaxonRows {
    std::vector<uint64_t> id, parent;
    std::vector<uint8_t> valid, rank;
};

inline bool is_rust_ws_tail(std::string_view s, size_t& cut) {
    size_t n = s.size();
    unsigned char c = (unsigned char)s[n - 1];
    if ((c >= 0x09 && c <= 0x0D) || c == 0x20) { cut = 1; return true; }
    if (n >= 2 && (unsigned char)s[n - 2] == 0xC2 && (c == 0x85 || c == 0xA0)) { cut = 2; return true; }
    if (n >= 3) {
        unsigned char a = (unsigned char)s[n - 3], b = (unsigned char)s[n - 2];
        bool ws = (a == 0xE1 && b == 0x9A && c == 0x80) || (a == 0xE2 && b == 0x80 && c >= 0x80 && c <= 0x8A) ||
                  (a == 0xE2 && b == 0x80 && (c == 0xA8 || c == 0xA9 || c == 0xAF)) || (a == 0xE2 && b == 0x81 && c == 0x9F) ||
                  (a == 0xE3 && b == 0x80 && c == 0x80);
        if (ws) { cut = 3; return true; }
    }
    return false;
}

inline void parse_taxon_line(std::string_view line, TaxonRows& rows) {
    size_t cut;
    while (!line.empty() && is_rust_ws_tail(line, cut)) line.remove_suffix(cut);  // trim_end()
    std::string_view f[5];
    size_t nf = 0, start = 0;
    for (;;) {
        size_t p = line.find('\t', start);
        if (nf < 5) f[nf] = line.substr(start, p == std::string_view::npos ? std::string_view::npos : p - start);
        nf++;
        if (p == std::string_view::npos) break;
        start = p + 1;
    }
    if (nf != 5) throw CliError("Taxon requires five fields");
    // all parses are evaluated, then the match arms in order: id error, rank error, parent error, valid byte
    std::optional<CliError> e_id, e_rank, e_parent;
    uint64_t id = 0, parent = 0;
    int rank = -1;
    try { id = parse_usize(f[0]); } catch (CliError& e) { e_id = e; }
    for (int r = 0; r < 32; r++) if (f[2] == RANK_NAMES[r]) { rank = r; break; }
    if (rank < 0) e_rank = CliError("Matching variant not found");
    try { parent = parse_usize(f[3]); } catch (CliError& e) { e_parent = e; }
    bool v1 = f[4] == std::string_view("\x01", 1), v0 = f[4] == std::string_view("\x00", 1);
    if (!e_id && !e_rank && !e_parent && (v1 || v0)) {
        rows.id.push_back(id); rows.parent.push_back(parent); rows.valid.push_back(v1 ? 1 : 0); rows.rank.push_back((uint8_t)rank);
        return;
    }
    if (e_id) throw *e_id;
    if (e_rank) throw *e_rank;
    if (e_parent) throw *e_parent;
    throw CliError("Couldn't parse the valid byte");
}

inline TaxonRows read_taxa_file(const std::string& path) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) {
        CliError e("Failed opening taxon file.");
        e.causes.push_back(std::string(strerror(errno)) + " (os error " + std::to_string(errno) + ")");
        throw e;
    }
    TaxonRows rows;
    LineSource ls(f);
    std::string_view l;
    for (;;) {
        bool more;
        try { more = ls.next(l); } catch (CliError&) { fclose(f); throw CliError("Failed to read all lines."); }
        if (!more) break;
        try { parse_taxon_line(l, rows); } catch (...) { fclose(f); throw; }
    }
    fclose(f);
    return rows;
}

}  // namespace umgap_host
