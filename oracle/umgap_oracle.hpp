// ============================================================================
// umgap_oracle.hpp — CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
//
// A literal, single-threaded C++17 restatement of the reference algorithm for
// the pept2lca -> taxa2agg hot path of unipept/umgap v1.1.1 (Rust).  It exists
// so that the HIP product path can be checked for bit-identical results.  Only
// tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build,
// link, import or execute anything in this directory; the product
// (umgap_amd/, include/) never does.
//
// Parity pin: the reference cannot be compiled here (Rust; no cargo/rustc in
// the image).  This restatement is pinned by every golden vector the
// reference's own unit tests hold for the path (tests/test_oracle_golden.py,
// list in DESIGN.md).  The index lookup itself lives in the third-party crate
// `fst` 0.3.5 whose source is absent and which no reference test exercises:
// that seam is "parity unpinned" and is restated from its contract only
// (exact-match ordered map: get(k) = v iff (k, v) was inserted).
//
// All `file:line` citations are relative to /root/reference/.
// Where the reference iterates a std::collections::HashMap/HashSet (random
// order), the oracle returns the SET of results reachable under any order.
// ============================================================================
#pragma once
#include <algorithm>
#include <atomic>
#include <thread>
#include <cstdint>
#include <cstring>
#include <deque>
#include <fstream>
#include <functional>
#include <istream>
#include <map>
#include <memory>
#include <optional>
#include <ostream>
#include <set>
#include <sstream>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

namespace umgap_oracle {

typedef size_t TaxonId;  // src/taxon.rs:18  (usize)
typedef size_t Depth;    // src/taxon.rs:305

// ---------------------------------------------------------------------------
// errors (src/errors.rs:16-37, src/taxon.rs:394-406, src/agg/mod.rs:46-57,
// src/commands/taxa2agg.rs:241-254).  `Error` => "Error: <display>" exit 1;
// `Panic` => Rust panic, exit 101 (src/main.rs:8 quick_main!).
// ---------------------------------------------------------------------------
enum class ErrKind {
    Msg, Io, ParseInt, ParseFloat, UnknownRank, UnknownTaxon, EmptyInput,
    InvalidInvocation, ParseMethod, ParseStrategy, Index
};

struct Error : std::exception {
    ErrKind kind;
    std::string msg;                  // Display of the error kind
    TaxonId taxon = 0;                // payload of UnknownTaxon
    std::vector<std::string> causes;  // chain_err causes ("Caused by: ...")
    Error(ErrKind k, std::string m, TaxonId t = 0) : kind(k), msg(std::move(m)), taxon(t) {}
    const char* what() const noexcept override { return msg.c_str(); }
};
struct Panic : std::exception {
    std::string msg;
    explicit Panic(std::string m) : msg(std::move(m)) {}
    const char* what() const noexcept override { return msg.c_str(); }
};

inline Error unknown_taxon(TaxonId t) {  // src/taxon.rs:400-404
    return Error(ErrKind::UnknownTaxon, "Unknown Taxon ID: " + std::to_string(t), t);
}
inline Error empty_input() {  // src/agg/mod.rs:51-55 (typo is the reference's)
    return Error(ErrKind::EmptyInput, "Aggregration called on an empty list");
}

// ---------------------------------------------------------------------------
// Rust std parsing rules the reference relies on.
// ---------------------------------------------------------------------------
// str::parse::<usize>() (core::num, radix 10): optional single leading '+',
// then >=1 ASCII digits; anything else is an error; overflow is an error.
inline size_t parse_usize(const std::string& s) {
    if (s.empty()) throw Error(ErrKind::ParseInt, "cannot parse integer from empty string");
    size_t i = 0;
    if (s[0] == '+' || s[0] == '-') {
        if (s.size() == 1) throw Error(ErrKind::ParseInt, "invalid digit found in string");
        if (s[0] == '-') throw Error(ErrKind::ParseInt, "invalid digit found in string");
        i = 1;
    }
    unsigned __int128 v = 0;
    for (; i < s.size(); i++) {
        char c = s[i];
        if (c < '0' || c > '9') throw Error(ErrKind::ParseInt, "invalid digit found in string");
        v = v * 10 + (unsigned)(c - '0');
        if (v > (unsigned __int128)UINT64_MAX)
            throw Error(ErrKind::ParseInt, "number too large to fit in target type");
    }
    return (size_t)v;
}

// str::parse::<f32>() (core::num::dec2flt grammar): [+-]? ( "inf" | "infinity"
// | "nan" (any case) | digits [ "." digits? ] | "." digits ) [ (e|E) [+-]? digits ]
inline float parse_f32(const std::string& s) {
    auto bad = []() { return Error(ErrKind::ParseFloat, "invalid float literal"); };
    if (s.empty()) throw Error(ErrKind::ParseFloat, "cannot parse float from empty string");
    size_t i = 0;
    if (s[i] == '+' || s[i] == '-') i++;
    std::string rest = s.substr(i);
    std::string low = rest;
    for (auto& c : low) c = (char)tolower((unsigned char)c);
    if (low == "inf" || low == "infinity" || low == "nan") {
        float v = (low == "nan") ? __builtin_nanf("") : __builtin_inff();
        return s[0] == '-' ? -v : v;
    }
    size_t j = 0, nd = 0;
    while (j < rest.size() && isdigit((unsigned char)rest[j])) { j++; nd++; }
    if (j < rest.size() && rest[j] == '.') {
        j++;
        while (j < rest.size() && isdigit((unsigned char)rest[j])) { j++; nd++; }
    }
    if (nd == 0) throw bad();
    if (j < rest.size() && (rest[j] == 'e' || rest[j] == 'E')) {
        j++;
        if (j < rest.size() && (rest[j] == '+' || rest[j] == '-')) j++;
        size_t ne = 0;
        while (j < rest.size() && isdigit((unsigned char)rest[j])) { j++; ne++; }
        if (ne == 0) throw bad();
    }
    if (j != rest.size()) throw bad();
    return strtof(s.c_str(), nullptr);  // correctly rounded, like dec2flt
}

// UTF-8 validity as enforced by BufRead::lines() (String::from_utf8).
inline bool valid_utf8(const std::string& s) {
    const unsigned char* p = (const unsigned char*)s.data();
    size_t n = s.size(), i = 0;
    while (i < n) {
        unsigned char c = p[i];
        if (c < 0x80) { i++; continue; }
        size_t len; uint32_t cp, min;
        if ((c & 0xE0) == 0xC0) { len = 2; cp = c & 0x1F; min = 0x80; }
        else if ((c & 0xF0) == 0xE0) { len = 3; cp = c & 0x0F; min = 0x800; }
        else if ((c & 0xF8) == 0xF0) { len = 4; cp = c & 0x07; min = 0x10000; }
        else return false;
        if (i + len > n) return false;
        for (size_t k = 1; k < len; k++) {
            if ((p[i + k] & 0xC0) != 0x80) return false;
            cp = (cp << 6) | (p[i + k] & 0x3F);
        }
        if (cp < min || cp > 0x10FFFF || (cp >= 0xD800 && cp <= 0xDFFF)) return false;
        i += len;
    }
    return true;
}

// str::trim_end(): strips trailing chars with the Unicode White_Space property.
inline std::string trim_end(const std::string& s) {
    size_t n = s.size();
    for (;;) {
        if (n == 0) break;
        unsigned char c = (unsigned char)s[n - 1];
        if ((c >= 0x09 && c <= 0x0D) || c == 0x20) { n--; continue; }
        // U+0085, U+00A0
        if (n >= 2 && (unsigned char)s[n - 2] == 0xC2 && (c == 0x85 || c == 0xA0)) { n -= 2; continue; }
        if (n >= 3) {
            unsigned char a = (unsigned char)s[n - 3], b = (unsigned char)s[n - 2];
            bool ws = false;
            if (a == 0xE1 && b == 0x9A && c == 0x80) ws = true;                  // U+1680
            if (a == 0xE2 && b == 0x80 && c >= 0x80 && c <= 0x8A) ws = true;     // U+2000-200A
            if (a == 0xE2 && b == 0x80 && (c == 0xA8 || c == 0xA9 || c == 0xAF)) ws = true;  // 2028/9/F
            if (a == 0xE2 && b == 0x81 && c == 0x9F) ws = true;                  // U+205F
            if (a == 0xE3 && b == 0x80 && c == 0x80) ws = true;                  // U+3000
            if (ws) { n -= 3; continue; }
        }
        break;
    }
    return s.substr(0, n);
}

// std::io::BufRead::lines(): split on '\n', strip one trailing "\r"; a final
// line without '\n' is still yielded; invalid UTF-8 is an InvalidData error.
struct LineReader {
    std::istream& in;
    explicit LineReader(std::istream& i) : in(i) {}
    // returns false at EOF; throws Error(Io) on invalid UTF-8
    bool next(std::string& line) {
        if (!std::getline(in, line)) return false;
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (!valid_utf8(line)) throw Error(ErrKind::Io, "stream did not contain valid UTF-8");
        return true;
    }
};

// ---------------------------------------------------------------------------
// rank (src/rank.rs:9-125)
// ---------------------------------------------------------------------------
static const char* const RANK_NAMES[32] = {  // src/rank.rs:10-41 (strum serialize)
    "no rank", "superkingdom", "domain", "realm", "kingdom", "subkingdom", "superphylum",
    "phylum", "subphylum", "superclass", "class", "subclass", "infraclass", "superorder",
    "order", "suborder", "infraorder", "parvorder", "superfamily", "family", "subfamily",
    "tribe", "subtribe", "genus", "subgenus", "species group", "species subgroup", "species",
    "subspecies", "varietas", "forma", "strain"};
enum : int { RANK_NO_RANK = 0, RANK_SUPERKINGDOM = 1, RANK_DOMAIN = 2, RANK_REALM = 3,
             RANK_SUPERPHYLUM = 6, RANK_SUPERCLASS = 9, RANK_SUPERORDER = 13,
             RANK_SUPERFAMILY = 18, RANK_TRIBE = 21, RANK_GENUS = 23, RANK_SPECIES_GROUP = 25,
             RANK_SPECIES = 27, RANK_COUNT = 32 };

inline int parse_rank(const std::string& s) {  // EnumString, src/rank.rs:8
    for (int i = 0; i < 32; i++)
        if (s == RANK_NAMES[i]) return i;
    throw Error(ErrKind::UnknownRank, "Matching variant not found");
}
// src/rank.rs:86-99.  `self < &Rank::X` goes through PartialOrd (:113-121) which is None when
// either side is NoRank, so every comparison is false for NoRank => falls to `else None`.
inline std::optional<size_t> rank_score(int r) {
    if (r == RANK_NO_RANK) return std::nullopt;
    // NB reference order of tests is as written: first match wins.
    if (r < RANK_SPECIES) {
        // `if self < &Rank::Species { Some(12) }` is the FIRST arm: every named rank above species
        // scores 12; the remaining arms are only reachable for ranks >= species.
        return 12;
    }
    if (r < RANK_SPECIES_GROUP) return 11;
    if (r < RANK_GENUS) return 10;
    if (r < RANK_TRIBE) return 9;
    if (r < RANK_SUPERFAMILY) return 8;
    if (r < RANK_SUPERORDER) return 7;
    if (r < RANK_SUPERCLASS) return 6;
    if (r < RANK_SUPERPHYLUM) return 5;
    if (r < RANK_REALM) return 4;
    if (r < RANK_DOMAIN) return 3;
    if (r < RANK_SUPERKINGDOM) return 2;
    return std::nullopt;
}

// ---------------------------------------------------------------------------
// taxon (src/taxon.rs)
// ---------------------------------------------------------------------------
struct Taxon {  // src/taxon.rs:22-33
    TaxonId id;
    std::string name;
    int rank;
    TaxonId parent;
    bool valid;
    bool operator==(const Taxon& o) const {
        return id == o.id && name == o.name && rank == o.rank && parent == o.parent && valid == o.valid;
    }
};

// impl FromStr for Taxon — src/taxon.rs:65-114
inline Taxon parse_taxon(const std::string& s) {
    std::string t = trim_end(s);
    std::vector<std::string> split;
    {
        size_t start = 0;
        for (;;) {
            size_t p = t.find('\t', start);
            if (p == std::string::npos) { split.push_back(t.substr(start)); break; }
            split.push_back(t.substr(start, p - start));
            start = p + 1;
        }
    }
    if (split.size() != 5) throw Error(ErrKind::Msg, "Taxon requires five fields");
    // match tuple: all four parses are evaluated, then arms tested in order (:99-113)
    std::optional<Error> e_id, e_rank, e_parent;
    TaxonId id = 0, parent = 0;
    int rank = 0;
    try { id = parse_usize(split[0]); } catch (Error& e) { e_id = e; }
    try { rank = parse_rank(split[2]); } catch (Error& e) { e_rank = e; }
    try { parent = parse_usize(split[3]); } catch (Error& e) { e_parent = e; }
    if (!e_id && !e_rank && !e_parent && split[4] == std::string("\x01", 1))
        return Taxon{id, split[1], rank, parent, true};
    if (!e_id && !e_rank && !e_parent && split[4] == std::string("\x00", 1))
        return Taxon{id, split[1], rank, parent, false};
    if (e_id) throw *e_id;          // InvalidID
    if (e_rank) throw *e_rank;      // UnknownRank
    if (e_parent) throw *e_parent;  // InvalidID
    throw Error(ErrKind::Msg, "Couldn't parse the valid byte");
}

// src/taxon.rs:119-128
inline std::vector<Taxon> read_taxa_stream(std::istream& in) {
    std::vector<Taxon> taxa;
    LineReader lr(in);
    std::string line;
    for (;;) {
        bool ok;
        try { ok = lr.next(line); } catch (Error&) { throw Error(ErrKind::Msg, "Failed to read all lines."); }
        if (!ok) break;
        taxa.push_back(parse_taxon(line));
    }
    return taxa;
}
inline std::vector<Taxon> read_taxa_file(const std::string& filename) {
    std::ifstream f(filename, std::ios::binary);
    if (!f) {
        Error e(ErrKind::Msg, "Failed opening taxon file.");
        e.causes.push_back(std::string(strerror(errno)));
        throw e;
    }
    return read_taxa_stream(f);
}

// TaxonList — src/taxon.rs:131-210
struct TaxonList {
    std::vector<std::optional<Taxon>> v;
    TaxonList() {}
    explicit TaxonList(const std::vector<Taxon>& taxons) {  // :135-145
        TaxonId max_id = 0;
        for (auto& t : taxons) max_id = std::max(max_id, t.id);
        v.resize(max_id + 1);
        for (auto& t : taxons) v[t.id] = t;
    }
    static TaxonList new_with_unknown(const std::vector<Taxon>& taxa, bool with_unknown) {  // :149-155
        TaxonList l(taxa);
        if (with_unknown && !l.v[0]) l.v[0] = Taxon{0, "unknown", RANK_NO_RANK, 0, false};
        return l;
    }
    std::vector<std::optional<TaxonId>> ancestry() const {  // :158-163
        std::vector<std::optional<TaxonId>> a(v.size());
        for (size_t i = 0; i < v.size(); i++)
            if (v[i]) a[i] = v[i]->parent;
        return a;
    }
    const Taxon* get(TaxonId i) const {  // :166-172
        if (i >= v.size()) return nullptr;
        return v[i] ? &*v[i] : nullptr;
    }
    const Taxon& get_or_unknown(TaxonId i) const {  // :176-178
        const Taxon* t = get(i);
        if (!t) throw unknown_taxon(i);
        return *t;
    }
    std::optional<size_t> score(TaxonId index) const {  // :181-191
        TaxonId current = index;
        while (const Taxon* t = get(current)) {
            if (t->parent == current || t->rank != RANK_NO_RANK) return rank_score(t->rank);
            current = t->parent;
        }
        return std::nullopt;
    }
    // :194-209 — returns taxon per rank index (nullptr if none)
    std::vector<const Taxon*> lineage(TaxonId index) const {
        std::vector<const Taxon*> arr(RANK_COUNT, nullptr);
        std::optional<TaxonId> next_id = index, prev_id;
        while (next_id != prev_id) {
            const Taxon& taxon = get_or_unknown(*next_id);
            if (taxon.rank != RANK_NO_RANK) arr[taxon.rank] = &taxon;
            prev_id = next_id;
            next_id = taxon.parent;
        }
        return arr;
    }
};

// TaxonTree — src/taxon.rs:214-302
struct TaxonTree {
    TaxonId root = 0;
    std::unordered_map<TaxonId, std::vector<TaxonId>> children;  // values keep TSV order
    TaxonId max = 0;

    TaxonTree() {}
    explicit TaxonTree(const std::vector<Taxon>& taxons) {  // :224-247
        if (taxons.empty()) throw Panic("index out of bounds: the len is 0 but the index is 0");
        children.reserve(taxons.size());
        max = taxons[0].id;
        std::set<TaxonId> roots;
        for (auto& t : taxons) roots.insert(t.id);
        for (auto& t : taxons) {
            if (t.id > max) max = t.id;
            if (t.id == t.parent) continue;
            children[t.parent].push_back(t.id);
            roots.erase(t.id);
        }
        if (roots.size() > 1) throw Panic("More than one root!");
        if (roots.empty()) throw Panic("There's no root!");
        root = *roots.begin();
    }
    size_t child_count(TaxonId whose) const {  // :284-286
        auto it = children.find(whose);
        return it == children.end() ? 0 : it->second.size();
    }
    // with_filtered (:251-271) is a preorder recursion; restated with an explicit stack that
    // performs the assignments in the same order (matters only if an id is reachable twice).
    // An id > max reachable from the root panics on `ancestors[current]` like the reference.
    template <class F>
    std::vector<std::optional<TaxonId>> filter_ancestors(F filter) const {  // :274-281
        std::vector<std::optional<TaxonId>> anc(max + 1);
        struct Frame { TaxonId node; std::optional<TaxonId> ancestor; };
        std::vector<Frame> stack;
        stack.push_back({root, std::optional<TaxonId>(root)});
        while (!stack.empty()) {
            Frame f = stack.back();
            stack.pop_back();
            std::optional<TaxonId> a = filter(f.node) ? std::optional<TaxonId>(f.node) : f.ancestor;
            if (f.node >= anc.size()) throw Panic("index out of bounds (filter_ancestors)");
            anc[f.node] = a;
            auto it = children.find(f.node);
            if (it != children.end())
                for (auto c = it->second.rbegin(); c != it->second.rend(); ++c) stack.push_back({*c, a});
        }
        return anc;
    }
    std::vector<std::optional<TaxonId>> snapping(const TaxonList& taxons, bool ranked_only) const {  // :294-301
        return filter_ancestors([&](TaxonId i) {
            const Taxon* t = taxons.get(i);
            return t ? (t->valid && (!ranked_only || t->rank != RANK_NO_RANK)) : false;
        });
    }
};

// EulerIterator — src/taxon.rs:309-392, restated state machine.
struct EulerIterator {
    const TaxonTree& tree;
    struct P { TaxonId node; size_t currentn; size_t children; };
    std::vector<P> path;
    TaxonId current;
    size_t currentn, children;
    explicit EulerIterator(const TaxonTree& t)
        : tree(t), current(t.root), currentn(0), children(t.child_count(t.root)) {}
    bool next(TaxonId& out_taxon, Depth& out_depth) {
        for (;;) {
            if (currentn > children) {  // Greater (:344-353)
                if (path.empty()) return false;
                P p = path.back(); path.pop_back();
                current = p.node; currentn = p.currentn; children = p.children;
                continue;
            } else if (currentn == children) {  // Equal (:354-370)
                TaxonId cur = current;
                if (path.empty()) {
                    current = 0; currentn = 1; children = 0;
                    out_taxon = tree.root; out_depth = 0;
                    return true;
                }
                P p = path.back(); path.pop_back();
                current = p.node; currentn = p.currentn; children = p.children;
                out_taxon = cur; out_depth = path.size() + 1;
                return true;
            } else {  // Less (:371-381)
                TaxonId cur = current;
                const std::vector<TaxonId>& ch = tree.children.at(cur);
                path.push_back({current, currentn + 1, children});
                current = ch[currentn];
                currentn = 0;
                children = tree.child_count(current);
                out_taxon = cur; out_depth = path.size() - 1;
                return true;
            }
        }
    }
};
inline std::vector<std::pair<TaxonId, Depth>> euler_tour(const TaxonTree& t) {
    std::vector<std::pair<TaxonId, Depth>> out;
    EulerIterator it(t);
    TaxonId a; Depth d;
    while (it.next(a, d)) out.emplace_back(a, d);
    return out;
}

// ---------------------------------------------------------------------------
// agg (src/agg/mod.rs)
// ---------------------------------------------------------------------------
// HashMap<TaxonId, f32> with the iteration order abstracted away: insertion order is kept so
// the f32 accumulation in count() is exactly the reference's (:32-34, input order per key).
struct Counts {
    std::vector<std::pair<TaxonId, float>> items;
    std::unordered_map<TaxonId, size_t> pos;
    bool empty() const { return items.empty(); }
    size_t size() const { return items.size(); }
    const float* get(TaxonId t) const {
        auto it = pos.find(t);
        return it == pos.end() ? nullptr : &items[it->second].second;
    }
    float get_or(TaxonId t, float d) const { const float* p = get(t); return p ? *p : d; }
    void add(TaxonId t, float c) {
        auto it = pos.find(t);
        if (it == pos.end()) { pos[t] = items.size(); items.emplace_back(t, 0.0f); it = pos.find(t); }
        items[it->second].second += c;
    }
};
inline Counts count(const std::vector<std::pair<TaxonId, float>>& taxons) {  // :27-36
    Counts c;
    for (auto& p : taxons) c.add(p.first, p.second);
    return c;
}
inline Counts filter(const Counts& freq, float lower_bound) {  // :39-44
    Counts out;
    for (auto& p : freq.items)
        if (p.second >= lower_bound) out.add(p.first, p.second);
    return out;
}

// Result of an aggregation under every HashMap/HashSet order the reference could see.
struct AggOutcome {
    std::set<TaxonId> ok;        // possible Ok(tid)
    std::set<TaxonId> unknown;   // possible Err(UnknownTaxon(tid))
    bool empty_input = false;    // Err(EmptyInput)
    bool may_panic = false;      // index out of bounds (exit 101)
    bool is_ok() const { return unknown.empty() && !empty_input && !may_panic; }
};

struct Aggregator {  // trait Aggregator, src/agg/mod.rs:12-21
    virtual ~Aggregator() {}
    virtual AggOutcome aggregate(const Counts& taxons) const = 0;
    AggOutcome counting_aggregate(const std::vector<TaxonId>& taxons) const {  // :17-20
        std::vector<std::pair<TaxonId, float>> v;
        for (auto t : taxons) v.emplace_back(t, 1.0f);
        return aggregate(count(v));
    }
};

// ---------------------------------------------------------------------------
// tree (src/tree/mod.rs, lca.rs, mix.rs)
// ---------------------------------------------------------------------------
struct Tree {  // src/tree/mod.rs:14-21, T = f32
    TaxonId root = 0;
    float value = 0.0f;
    std::vector<Tree> children;  // order = HashSet iteration order in the reference (random)
};

// Tree::new (:29-48).  Errors are order dependent (BFS over HashMap keys); all reachable error
// payloads are collected into `out` and false is returned when any exists.
inline bool tree_new(TaxonId root, const std::vector<std::optional<TaxonId>>& parents,
                     const Counts& taxons, Tree& result, AggOutcome& out) {
    std::map<TaxonId, std::set<TaxonId>> tree;
    // The reference's queue holds the keys (random order) and then newly discovered parents, so
    // nodes are popped level by level (level = distance from the nearest key at discovery).  The
    // first failing pop aborts: the possible failures are exactly those of the first level that
    // has any.  Within a level the order is random => every failure of that level is possible.
    std::vector<TaxonId> level;
    for (auto& p : taxons.items) level.push_back(p.first);
    while (!level.empty()) {
        bool failed = false;
        for (TaxonId id : level) {
            if (id >= parents.size()) { out.may_panic = true; failed = true; }      // parents[id] OOB
            else if (!parents[id]) { out.unknown.insert(id); failed = true; }       // :38
        }
        if (failed) return false;
        std::vector<TaxonId> next;
        for (TaxonId id : level) {
            TaxonId parent = *parents[id];
            if (id == parent) continue;
            if (!tree.count(parent)) next.push_back(parent);
            tree[parent].insert(id);
        }
        level.swap(next);
    }
    std::function<Tree(TaxonId)> create = [&](TaxonId r) {  // :50-68
        Tree t;
        t.root = r;
        t.value = taxons.get_or(r, 0.0f);
        auto it = tree.find(r);
        if (it != tree.end())
            for (TaxonId c : it->second) t.children.push_back(create(c));
        return t;
    };
    result = create(root);
    return true;
}
inline Tree tree_collapse(const Tree& self) {  // :71-86, combine = Add::add
    float value = self.value;
    const Tree* n = &self;
    while (n->children.size() == 1) {
        n = &n->children[0];
        value = value + n->value;
    }
    Tree t;
    t.root = n->root;
    t.value = value;
    for (auto& c : n->children) t.children.push_back(tree_collapse(c));
    return t;
}
inline Tree tree_aggregate(const Tree& self) {  // :90-101 (fold order = child order; exact for ints)
    Tree t;
    t.root = self.root;
    for (auto& c : self.children) t.children.push_back(tree_aggregate(c));
    float value = self.value;
    for (auto& c : t.children) value = value + c.value;
    t.value = value;
    return t;
}

struct TreeLCACalculator : Aggregator {  // src/tree/lca.rs:11-41
    TaxonId root;
    std::vector<std::optional<TaxonId>> parents;
    TreeLCACalculator(TaxonId r, const TaxonList& taxonomy) : root(r), parents(taxonomy.ancestry()) {}
    AggOutcome aggregate(const Counts& taxons) const override {
        AggOutcome out;
        if (taxons.empty()) { out.empty_input = true; return out; }  // :33-35
        Tree t;
        if (!tree_new(root, parents, taxons, t, out)) return out;
        out.ok.insert(tree_collapse(t).root);  // :36-37
        return out;
    }
};

struct TreeMixCalculator : Aggregator {  // src/tree/mix.rs:16-65
    TaxonId root;
    std::vector<std::optional<TaxonId>> parents;
    float factor;
    TreeMixCalculator(TaxonId r, const TaxonList& taxonomy, float f)
        : root(r), parents(taxonomy.ancestry()), factor(f) {}
    AggOutcome aggregate(const Counts& taxons) const override {
        AggOutcome out;
        if (taxons.empty()) { out.empty_input = true; return out; }  // :45-47
        Tree t;
        if (!tree_new(root, parents, taxons, t, out)) return out;
        Tree subtree = tree_aggregate(tree_collapse(t));  // :48-50
        // :52-62 — max_by_key returns the LAST maximum in (random) child order: every child that
        // attains the maximum is a possible `max`; explore them all.
        std::function<void(const Tree*)> descend = [&](const Tree* base) {
            if (base->children.empty()) { out.ok.insert(base->root); return; }
            float mx = base->children[0].value;
            for (auto& c : base->children) if (c.value > mx) mx = c.value;
            if (mx / base->value < factor) { out.ok.insert(base->root); return; }  // :57-59
            for (auto& c : base->children)
                if (c.value == mx) descend(&c);
        };
        descend(&subtree);
        return out;
    }
};

// ---------------------------------------------------------------------------
// rmq (src/rmq/mod.rs) — usize is 64 bit, block size() = 64
// ---------------------------------------------------------------------------
struct RMQ {
    std::vector<size_t> array, block_min_;
    std::vector<std::vector<size_t>> sparse_;
    std::vector<size_t> labels_;

    static size_t clearbits(size_t n, size_t x) { return (n >> x) << x; }   // :26-28
    static size_t size() { return 64; }                                     // :30-32
    static size_t intlog2(size_t n) {                                       // :34-36
        size_t lz = n == 0 ? 64 : (size_t)__builtin_clzll(n);
        return lz ^ (size() - 1);
    }
    RMQ() {}
    explicit RMQ(std::vector<size_t> a) : array(std::move(a)) {  // :40-50
        block_min_ = block_min(array);
        sparse_ = sparse(array, block_min_);
        labels_ = labels(array);
    }
    static std::vector<size_t> block_min(const std::vector<size_t>& array) {  // :53-66
        std::vector<size_t> out;
        for (size_t i = 0; i * size() < array.size(); i++) {
            size_t lo = i * size(), hi = std::min(array.size(), lo + size());
            size_t best = lo;  // min_by_key returns the FIRST minimum
            for (size_t k = lo; k < hi; k++)
                if (array[k] < array[best]) best = k;
            out.push_back(best);
        }
        return out;
    }
    static std::vector<size_t> aggregate_minima(const std::vector<size_t>& array, size_t shift,
                                                const std::vector<size_t>& minima) {  // :68-74
        std::vector<size_t> out;
        for (size_t i = 0; i + shift < minima.size(); i++) {
            size_t l = minima[i], r = minima[i + shift];
            out.push_back(array[l] < array[r] ? l : r);
        }
        return out;
    }
    static std::vector<std::vector<size_t>> sparse(const std::vector<size_t>& array,
                                                   const std::vector<size_t>& block_min) {  // :77-86
        size_t length = intlog2(block_min.size());
        std::vector<std::vector<size_t>> sp;
        sp.push_back(aggregate_minima(array, 1, block_min));
        for (size_t i = 1; i < length; i++) sp.push_back(aggregate_minima(array, (size_t)1 << i, sp[i - 1]));
        return sp;
    }
    static std::vector<size_t> labels(const std::vector<size_t>& array) {  // :89-107
        std::vector<size_t> gstack, lab;
        for (size_t i = 0; i < array.size(); i++) {
            if (i % size() == 0) gstack.clear();
            lab.push_back(0);
            while (!gstack.empty() && array[i] < array[gstack.back()]) gstack.pop_back();
            if (!gstack.empty()) {
                size_t g = gstack.back();
                lab[i] = lab[g] | ((size_t)1 << (g % size()));
            }
            gstack.push_back(i);
        }
        return lab;
    }
    static size_t min_in_block(const std::vector<size_t>& labels, size_t left, size_t right) {  // :110-117
        size_t v = clearbits(labels.at(right), left % size());
        if (v == 0) return right;
        return clearbits(left, intlog2(size())) + (size_t)__builtin_ctzll(v);
    }
    size_t query(size_t start, size_t end) const {  // :121-156
        if (start == end) return start;
        size_t left = start < end ? start : end, right = start < end ? end : start;
        size_t log2 = intlog2(size()), sz = size();
        size_t block_diff = (right >> log2) - (left >> log2);
        if (block_diff == 0) return min_in_block(labels_, left, right);
        size_t l = min_in_block(labels_, left, clearbits(left, log2) + sz - 1);
        size_t r = min_in_block(labels_, clearbits(right, log2), right);
        if (block_diff == 1) return array.at(l) <= array.at(r) ? l : r;
        size_t m;
        if (block_diff == 2) {
            m = block_min_.at((left >> log2) + 1);
        } else {
            size_t k = intlog2(block_diff - 1) - 1;
            size_t t1 = sparse_.at(k).at((left >> log2) + 1);
            size_t t2 = sparse_.at(k).at((right >> log2) - ((size_t)1 << (k + 1)));
            m = array.at(t1) <= array.at(t2) ? t1 : t2;
        }
        size_t ex = array.at(l) <= array.at(m) ? l : m;
        return array.at(ex) <= array.at(r) ? ex : r;
    }
};

// rmq::lca::LCACalculator — src/rmq/lca.rs:11-91
struct RmqLCACalculator : Aggregator {
    std::unordered_map<TaxonId, size_t> first_occurences;
    std::vector<TaxonId> euler;
    RMQ rmq_info;
    explicit RmqLCACalculator(const TaxonTree& taxonomy) {  // :23-39
        std::vector<size_t> depths;
        size_t i = 0;
        for (auto& p : euler_tour(taxonomy)) {
            euler.push_back(p.first);
            depths.push_back(p.second);
            first_occurences.emplace(p.first, i);  // or_insert
            i++;
        }
        rmq_info = RMQ(depths);
    }
    size_t first_occurence(TaxonId t) const {  // :49-54
        auto it = first_occurences.find(t);
        if (it == first_occurences.end()) throw unknown_taxon(t);
        return it->second;
    }
    TaxonId lca(TaxonId left, TaxonId right) const {  // :42-47
        size_t li = first_occurence(left), ri = first_occurence(right);
        return euler[rmq_info.query(li, ri)];
    }
    // :60-90 for ONE given key order
    TaxonId aggregate_order(const std::vector<TaxonId>& keys) const {
        std::vector<size_t> indices;
        for (auto t : keys) indices.push_back(first_occurence(t));
        size_t consensus = indices[0];
        std::optional<size_t> join_level;
        for (size_t n = 1; n < indices.size(); n++) {
            size_t next = indices[n];
            if (consensus == next) continue;
            size_t rmq = rmq_info.query(consensus, next);
            size_t lca;
            std::optional<size_t> level;
            bool a = (rmq == consensus), b = (rmq == next);
            if (!a && !b) { lca = rmq; level = rmq_info.array[rmq]; }
            else if (a && !b) { lca = next; level = join_level; }
            else if (!a && b) { lca = consensus; level = join_level; }
            else throw Panic("Impossibru!");
            if (join_level && rmq_info.array[lca] > *join_level) lca = rmq;
            consensus = lca;
            join_level = level;
        }
        return euler[consensus];
    }
    AggOutcome aggregate(const Counts& taxons) const override {
        AggOutcome out;
        if (taxons.empty()) { out.empty_input = true; return out; }
        std::vector<TaxonId> keys;
        for (auto& p : taxons.items) {
            if (!first_occurences.count(p.first)) out.unknown.insert(p.first);
            keys.push_back(p.first);
        }
        if (!out.unknown.empty()) return out;
        // HashMap key order is random: evaluate insertion order, its reverse, sorted orders and a
        // set of deterministic shuffles (all n! orders when n <= 6).
        std::vector<TaxonId> k = keys;
        std::sort(k.begin(), k.end());
        if (k.size() <= 6) {
            do { out.ok.insert(aggregate_order(k)); } while (std::next_permutation(k.begin(), k.end()));
        } else {
            out.ok.insert(aggregate_order(keys));
            out.ok.insert(aggregate_order(k));
            std::reverse(k.begin(), k.end());
            out.ok.insert(aggregate_order(k));
            uint64_t s = 0x9E3779B97F4A7C15ull ^ keys.size();
            for (int rep = 0; rep < 8; rep++) {
                for (size_t i = k.size() - 1; i > 0; i--) {
                    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
                    std::swap(k[i], k[s % (i + 1)]);
                }
                out.ok.insert(aggregate_order(k));
            }
        }
        return out;
    }
};

// rmq::rtl::RTLCalculator — src/rmq/rtl.rs:14-58
struct RTLCalculator : Aggregator {
    TaxonId root;
    std::vector<std::optional<TaxonId>> ancestors;
    RTLCalculator(TaxonId r, const TaxonList& taxons) : root(r), ancestors(taxons.ancestry()) {  // :27-31
        if (root >= ancestors.size()) throw Panic("index out of bounds (ancestors[root])");
        ancestors[root] = std::nullopt;
    }
    AggOutcome aggregate(const Counts& taxons) const override {  // :39-57
        AggOutcome out;
        std::vector<std::pair<TaxonId, float>> rtl = taxons.items;  // clone
        for (auto& tc : rtl) {
            TaxonId next = tc.first;
            bool panicked = false;
            for (;;) {
                if (next >= ancestors.size()) { panicked = true; break; }  // self.ancestors[next] OOB
                if (!ancestors[next]) break;
                TaxonId ancestor = *ancestors[next];
                tc.second += taxons.get_or(ancestor, 0.0f);
                next = ancestor;
            }
            if (panicked) { out.may_panic = true; continue; }
            if (next != root) out.unknown.insert(next);  // :47-49
        }
        if (out.may_panic || !out.unknown.empty()) return out;
        if (rtl.empty()) { out.empty_input = true; return out; }  // :55
        float mx = rtl[0].second;
        for (auto& tc : rtl) if (tc.second > mx) mx = tc.second;
        for (auto& tc : rtl) if (tc.second == mx) out.ok.insert(tc.first);  // max_by_key tie set
        return out;
    }
};

// rmq::mix::MixCalculator — src/rmq/mix.rs:16-94
struct RmqMixCalculator : Aggregator {
    RmqLCACalculator lca_aggregator;
    float factor;
    RmqMixCalculator(const TaxonTree& taxonomy, float f) : lca_aggregator(taxonomy), factor(f) {}
    AggOutcome aggregate(const Counts& taxons) const override {  // :57-93
        AggOutcome out;
        for (auto& p : taxons.items)
            if (!lca_aggregator.first_occurences.count(p.first)) out.unknown.insert(p.first);
        if (!out.unknown.empty()) return out;
        struct W { float lca = 0.0f, rtl = 0.0f; };
        std::map<TaxonId, W> weights;
        std::deque<TaxonId> queue;
        for (auto& p : taxons.items) queue.push_back(p.first);
        while (!queue.empty()) {
            TaxonId left = queue.front(); queue.pop_front();
            if (weights.count(left)) continue;
            for (auto& rc : taxons.items) {
                TaxonId right = rc.first;
                float cnt = rc.second;
                TaxonId l = lca_aggregator.lca(left, right);
                if (l == left || l == right) {
                    W& w = weights[left];
                    if (l == left) w.lca += cnt;
                    if (l == right) w.rtl += cnt;
                } else {
                    queue.push_back(l);
                }
            }
        }
        if (weights.empty()) { out.empty_input = true; return out; }  // :92
        auto factorize = [&](const W& w) -> float {  // :52-54, f32 arithmetic, no contraction
            volatile float a = w.lca * factor;
            volatile float one_minus = 1.0f - factor;
            volatile float b = w.rtl * one_minus;
            return a + b;
        };
        bool first = true;
        float mx = 0;
        for (auto& kv : weights) { float v = factorize(kv.second); if (first || v > mx) { mx = v; first = false; } }
        for (auto& kv : weights) if (factorize(kv.second) == mx) out.ok.insert(kv.first);
        return out;
    }
};

// ---------------------------------------------------------------------------
// io::fasta (src/io/fasta.rs)
// ---------------------------------------------------------------------------
struct Record {  // :78-84
    std::string header;
    std::vector<std::string> sequence;
};
struct FastaReader {  // :18-73
    LineReader lines;
    bool unwrap;
    bool have_peek = false, peek_eof = false, peek_err = false;
    std::string peek_line;
    Error peek_error{ErrKind::Io, ""};
    FastaReader(std::istream& in, bool unwrap_) : lines(in), unwrap(unwrap_) {}
    void fill() {
        if (have_peek) return;
        have_peek = true; peek_eof = false; peek_err = false;
        try { if (!lines.next(peek_line)) peek_eof = true; }
        catch (Error& e) { peek_err = true; peek_error = e; }
    }
    // Ok(None) => false; Ok(Some) => true; Err => throws
    bool read_record(Record& rec) {  // :38-67
        fill();
        if (peek_eof) return false;
        have_peek = false;
        if (peek_err) throw peek_error;
        std::string header = peek_line;
        if (header.empty() || header[0] != '>')
            throw Error(ErrKind::Io, "Expected > at beginning of fasta header.");  // :44-49
        header.erase(0, 1);
        std::vector<std::string> sequence;
        for (;;) {
            fill();
            if (peek_eof || peek_err) break;             // peek None / Err => loop ends (:53-59)
            if (!peek_line.empty() && peek_line[0] == '>') break;
            sequence.push_back(peek_line);
            have_peek = false;
        }
        if (unwrap) {  // :62-64
            std::string cat;
            for (auto& s : sequence) cat += s;
            sequence.clear();
            sequence.push_back(cat);
        }
        rec.header = header;
        rec.sequence = sequence;
        return true;
    }
};
static const size_t FASTA_WIDTH = 70;  // :15
struct FastaWriter {  // :141-181
    std::ostream& out;
    std::string separator;
    bool wrap;
    FastaWriter(std::ostream& o, std::string sep, bool w) : out(o), separator(std::move(sep)), wrap(w) {}
    void write_record(const Record& r) {  // :165-180
        out << '>' << r.header;
        std::string sequence;
        for (size_t i = 0; i < r.sequence.size(); i++) {
            if (i) sequence += separator;
            sequence += r.sequence[i];
        }
        if (!wrap) {
            out << '\n' << sequence;
        } else {
            for (size_t i = 0; i < sequence.size(); i += FASTA_WIDTH)
                out << '\n' << sequence.substr(i, FASTA_WIDTH);
        }
        if (!sequence.empty()) out << '\n';
    }
};

// ---------------------------------------------------------------------------
// The index seam: fst::Map::get(&[u8]) -> Option<u64> (crate fst 0.3.5, absent; calls at
// src/commands/pept2lca.rs:76,78,93).  Contract only: exact-match ordered map.
// ---------------------------------------------------------------------------
struct Index {
    virtual ~Index() {}
    virtual std::optional<uint64_t> get(const std::string& key) const = 0;
};
// Sorted-array index (keys kept as byte strings, binary search = log2 N dependent accesses);
// built from the TSV `umgap buildindex` consumes (src/commands/buildindex.rs:32-48): one
// "key<TAB>value" per line, keys in strictly increasing byte order (fst::MapBuilder::insert).
struct SortedIndex : Index {
    std::vector<std::string> keys;
    std::vector<uint64_t> vals;
    static SortedIndex from_tsv(std::istream& in) {
        SortedIndex ix;
        std::string line;
        while (std::getline(in, line)) {
            if (!line.empty() && line.back() == '\r') line.pop_back();
            if (line.empty()) continue;
            size_t tab = line.find('\t');
            if (tab == std::string::npos || line.find('\t', tab + 1) != std::string::npos)
                throw Error(ErrKind::Index, "CSV deserialize error: expected two fields");
            std::string k = line.substr(0, tab);
            std::string vs = line.substr(tab + 1);
            uint64_t v;
            try { v = parse_usize(vs); } catch (Error&) {
                throw Error(ErrKind::Index, "CSV deserialize error: invalid integer");
            }
            if (!ix.keys.empty() && !(ix.keys.back() < k)) {
                if (ix.keys.back() == k) throw Error(ErrKind::Index, "Error inserting duplicate key: " + k);
                throw Error(ErrKind::Index, "Error inserting out-of-order key: " + k);
            }
            ix.keys.push_back(k);
            ix.vals.push_back(v);
        }
        return ix;
    }
    std::optional<uint64_t> get(const std::string& key) const override {
        auto it = std::lower_bound(keys.begin(), keys.end(), key);
        if (it == keys.end() || *it != key) return std::nullopt;
        return vals[(size_t)(it - keys.begin())];
    }
};

// ---------------------------------------------------------------------------
// commands
// ---------------------------------------------------------------------------
// ChunkedRecords::next — src/io/fasta.rs:118-137: up to chunk_size records; a read error is
// returned AT ONCE and the records collected for this chunk are dropped with it.
inline bool next_chunk(FastaReader& reader, size_t chunk_size, std::vector<Record>& chunk) {
    chunk.clear();
    Record r;
    while (chunk.size() < chunk_size) {
        if (!reader.read_record(r)) break;  // throws on Err
        chunk.push_back(r);
    }
    return !chunk.empty();
}

// pept2lca record loop — src/commands/pept2lca.rs:81-101 with RAYON_NUM_THREADS=1 ordering
// (input order).  Output is built per chunk and printed when the chunk is done (:89,:98): an
// error or panic inside a chunk prints nothing of that chunk.
inline void pept2lca(std::istream& in, std::ostream& out, const Index& index, bool one_on_one, size_t chunk_size = 240) {
    std::optional<uint64_t> dflt;
    if (one_on_one) dflt = 0;  // :81
    FastaReader reader(in, false);
    std::vector<Record> chunk;
    while (next_chunk(reader, chunk_size, chunk)) {
        std::string chunk_output;
        for (auto& read : chunk) {
            chunk_output += ">" + read.header + "\n";  // :91
            for (auto& seq : read.sequence) {
                std::optional<uint64_t> lca = index.get(seq);
                if (!lca) lca = dflt;
                if (lca) chunk_output += std::to_string(*lca) + "\n";  // :93-95
            }
        }
        out << chunk_output;
    }
}

// prot2kmer2lca stream loop — src/commands/prot2kmer2lca.rs:150-193 (input order).
inline void prot2kmer2lca(std::istream& in, std::ostream& out, const Index& index, size_t k,
                          bool one_on_one, size_t chunk_size = 240) {
    std::optional<uint64_t> dflt;
    if (one_on_one) dflt = 0;
    FastaReader reader(in, true);
    std::vector<Record> chunk;
    while (next_chunk(reader, chunk_size, chunk)) {
        std::string chunk_output;
        for (auto& read : chunk) {
            // :172 — `read.sequence.get(0).filter(|p| p.len() >= k)`; shorter reads vanish entirely
            if (read.sequence.empty() || read.sequence[0].size() < k) continue;
            const std::string& prot = read.sequence[0];
            chunk_output += ">" + read.header + "\n";
            // NB `&prot[i..i + k]` panics if the slice cuts a multi-byte UTF-8 char (:176); the chunk's
            // output is then never written (:186-189)
            for (size_t i = 0; i + k <= prot.size(); i++) {
                if (((unsigned char)prot[i] & 0xC0) == 0x80 ||
                    (i + k < prot.size() && ((unsigned char)prot[i + k] & 0xC0) == 0x80))
                    throw Panic("byte index is not a char boundary");
                std::optional<uint64_t> lca = index.get(prot.substr(i, k));
                if (!lca) lca = dflt;
                if (lca) chunk_output += std::to_string(*lca) + "\n";
            }
        }
        out << chunk_output;
    }
}

// The same loop with the chunks looked up on `threads` threads, as the reference does with rayon's par_bridge
// (src/commands/prot2kmer2lca.rs:163-192).  The reference prints a chunk when it is done (completion order); this restatement
// keeps INPUT order (= the reference under RAYON_NUM_THREADS=1, the parity target), so its bytes are those of the loop above.
// Used by bench.py's cpu_baseline_end_to_end so that the CPU twin's lookup stage gets every core the reference's would.
inline void prot2kmer2lca_threads(std::istream& in, std::ostream& out, const Index& index, size_t k, bool one_on_one, size_t chunk_size,
                                  unsigned threads) {
    if (threads <= 1) { prot2kmer2lca(in, out, index, k, one_on_one, chunk_size); return; }
    std::optional<uint64_t> dflt;
    if (one_on_one) dflt = 0;
    FastaReader reader(in, true);
    const size_t batch = (size_t)threads * 8;
    std::vector<std::vector<Record>> chunks(batch);
    std::vector<std::string> outs(batch);
    std::vector<int> state(batch);          // 0 ok, 1 panic
    bool more = true;
    while (more) {
        size_t n = 0;
        std::exception_ptr read_error;
        try {
            while (n < batch && (more = next_chunk(reader, chunk_size, chunks[n]))) n++;
        } catch (...) { read_error = std::current_exception(); more = false; }   // an Err ends the iteration: the open chunk is dropped (src/io/fasta.rs:118-137)
        std::atomic<size_t> next{0};
        auto work = [&]() {
            for (size_t c; (c = next.fetch_add(1)) < n;) {
                std::string& o = outs[c];
                o.clear();
                state[c] = 0;
                try {
                    for (auto& read : chunks[c]) {
                        if (read.sequence.empty() || read.sequence[0].size() < k) continue;
                        const std::string& prot = read.sequence[0];
                        o += ">" + read.header + "\n";
                        for (size_t i = 0; i + k <= prot.size(); i++) {
                            if (((unsigned char)prot[i] & 0xC0) == 0x80 || (i + k < prot.size() && ((unsigned char)prot[i + k] & 0xC0) == 0x80))
                                throw Panic("byte index is not a char boundary");
                            std::optional<uint64_t> lca = index.get(prot.substr(i, k));
                            if (!lca) lca = dflt;
                            if (lca) o += std::to_string(*lca) + "\n";
                        }
                    }
                } catch (Panic&) { state[c] = 1; }
            }
        };
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < threads; t++) pool.emplace_back(work);
        work();
        for (auto& t : pool) t.join();
        for (size_t c = 0; c < n; c++) {
            if (state[c]) throw Panic("byte index is not a char boundary");
            out << outs[c];
        }
        if (read_error) std::rethrow_exception(read_error);
    }
}

// prot2tryp2lca — src/commands/prot2tryp2lca.rs:88-140 with the default cleavage pattern
// "([KR])([^P])" (:66-67).  The regex is applied twice with "$1\n$2" (:112-115); one pass of
// Regex::replace_all = leftmost non-overlapping matches, restated here character by character.
inline std::string tryp_replace_pass(const std::string& s) {
    std::string o;
    size_t i = 0;
    while (i < s.size()) {
        char c = s[i];
        if ((c == 'K' || c == 'R') && i + 1 < s.size() && s[i + 1] != 'P') {
            // [^P] matches one whole UTF-8 character
            size_t n = 1;
            unsigned char d = (unsigned char)s[i + 1];
            if (d >= 0xF0) n = 4; else if (d >= 0xE0) n = 3; else if (d >= 0xC0) n = 2;
            o.push_back(c);
            o.push_back('\n');
            o.append(s, i + 1, n);
            i += 1 + n;
        } else {
            o.push_back(c);
            i++;
        }
    }
    return o;
}
inline bool next_chunk(FastaReader& reader, size_t chunk_size, std::vector<Record>& chunk);
inline void prot2tryp2lca(std::istream& in, std::ostream& out_final, const Index& index, bool one_on_one, size_t min_length,
                          size_t max_length, const std::string& contains, const std::string& lacks, size_t chunk_size = 240) {
    std::optional<uint64_t> dflt;
    if (one_on_one) dflt = 0;
    std::set<char> cset(contains.begin(), contains.end()), lset(lacks.begin(), lacks.end());
    FastaReader reader(in, false);
    std::vector<Record> chunk;
    while (next_chunk(reader, chunk_size, chunk)) {
      std::ostringstream out;  // chunk_output (:106,:136)
      for (auto& read : chunk) {
        out << '>' << read.header << '\n';
        for (auto& seq : read.sequence) {
            std::string d = tryp_replace_pass(tryp_replace_pass(seq));
            for (auto& ch : d) if (ch == '*') ch = '\n';               // .replace("*", "\n")
            size_t st = 0;
            for (;;) {                                                   // .lines()
                size_t p = d.find('\n', st);
                std::string pep = d.substr(st, p == std::string::npos ? std::string::npos : p - st);
                if (!pep.empty() && pep.back() == '\r') pep.pop_back();
                bool ok = !pep.empty() && pep.size() >= min_length && pep.size() <= max_length;
                if (ok && !(cset.empty() && lset.empty())) {
                    std::set<char> have(pep.begin(), pep.end());
                    for (char c : cset) if (!have.count(c)) ok = false;
                    for (char c : lset) if (have.count(c)) ok = false;
                }
                if (ok) {
                    std::optional<uint64_t> lca = index.get(pep);
                    if (!lca) lca = dflt;
                    if (lca) out << *lca << '\n';
                }
                if (p == std::string::npos) break;
                st = p + 1;
            }
        }
      }
      out_final << out.str();
    }
}

// seedextend — src/commands/seedextend.rs:82-179
inline void seedextend(std::istream& in, std::ostream& out, size_t min_seed_size, size_t max_gap_size,
                       const TaxonList* by_id /* -r, from new_with_unknown(.., true) */, size_t penalty,
                       bool overflow_checks = false /* false: a release build, what `cargo install` ships */) {
    FastaWriter writer(out, "\n", false);
    FastaReader reader(in, false);
    Record record;
    while (reader.read_record(record)) {
        std::vector<TaxonId> taxons;
        for (auto& s : record.sequence) taxons.push_back(parse_usize(s));  // :94-98
        taxons.push_back(0);                                              // :99
        std::vector<std::pair<size_t, size_t>> seeds;
        size_t start = 0, end = 1;
        TaxonId last_tid = taxons[start];
        size_t same_tid = 1, same_max = 1;
        while (end < taxons.size()) {  // :107-142
            if (last_tid == taxons[end]) { same_tid += 1; end += 1; continue; }
            if (last_tid == 0 && same_tid > max_gap_size) {
                if (same_max >= min_seed_size) seeds.emplace_back(start, end - same_tid);
                start = end;
                last_tid = taxons[end];
                same_tid = 1;
                same_max = 1;
                end += 1;
                continue;
            }
            if (last_tid == 0 && (end - start) == same_tid) {
                end += 1;
                start = end;
                continue;
            }
            if (last_tid != 0) same_max = std::max(same_max, same_tid);
            last_tid = taxons[end];
            same_tid = 1;
            end += 1;
        }
        if (same_max >= min_seed_size) {  // :143-148
            if (last_tid == 0) end -= same_tid;
            seeds.emplace_back(start, end);
        }
        if (by_id) {  // :150-163 — max_by_key keeps the LAST maximum
            if (!seeds.empty()) {
                size_t best = 0, best_score = 0;
                bool first = true;
                for (size_t i = 0; i < seeds.size(); i++) {
                    // .skip(s).take(e - s): the scanner can push a seed with s > e (:119, a leading zero-run <= g, one taxon, a
                    // zero-run > g, with -s 1).  `e - s` then underflows: a build with overflow checks (debug) panics right here,
                    // a release build wraps, and take(2^64 - ..) covers every taxon from s on, the sentinel included.  The sum
                    // (`.sum::<usize>()`) wraps in a release build as well, like size_t here.
                    const size_t s0 = seeds[i].first, e0 = seeds[i].second;
                    if (s0 > e0 && overflow_checks) throw Panic("attempt to subtract with overflow");
                    const size_t stop = s0 > e0 ? taxons.size() : std::min(e0, taxons.size());
                    size_t sc = 0;
                    for (size_t j = s0; j < stop; j++) {
                        auto s = by_id->score(taxons[j]);
                        sc += s ? *s : penalty;
                    }
                    if (first || sc >= best_score) { best = i; best_score = sc; first = false; }
                }
                std::pair<size_t, size_t> keep = seeds[best];
                seeds.clear();
                seeds.push_back(keep);
            }
        }
        Record o;
        o.header = record.header;
        for (auto& se : seeds) {
            if (se.first > se.second || se.second > taxons.size()) throw Panic("slice index out of range");
            for (size_t j = se.first; j < se.second; j++) o.sequence.push_back(std::to_string(taxons[j]));
        }
        writer.write_record(o);
    }
}

// uniq — src/commands/uniq.rs:56-84
inline void uniq(std::istream& in, std::ostream& out, const std::string& separator, bool wrap,
                 const std::optional<std::string>& delimiter) {
    std::optional<Record> last;
    FastaWriter writer(out, separator, wrap);
    FastaReader reader(in, false);
    Record record;
    while (reader.read_record(record)) {
        if (delimiter) {
            size_t p = record.header.find(*delimiter);
            if (p != std::string::npos) record.header.resize(p);
        }
        if (last) {
            if (last->header == record.header) {
                last->sequence.insert(last->sequence.end(), record.sequence.begin(), record.sequence.end());
            } else {
                writer.write_record(*last);
                last = record;
            }
        } else {
            last = record;
        }
    }
    if (last) writer.write_record(*last);
}

// taxa2agg — src/commands/taxa2agg.rs:102-183
enum class Method { Tree, RangeMinimumQuery };                                  // :186-190
enum class Strategy { LowestCommonAncestor, Hybrid, MaximumRootToLeafPath };    // :214-219
inline Method parse_method(const std::string& s) {  // :199-209
    if (s == "tree") return Method::Tree;
    if (s == "rmq") return Method::RangeMinimumQuery;
    throw Error(ErrKind::ParseMethod, "Unparseable method: " + s);
}
inline Strategy parse_strategy(const std::string& s) {  // :228-239
    if (s == "lca*") return Strategy::LowestCommonAncestor;
    if (s == "hybrid") return Strategy::Hybrid;
    if (s == "mrtl") return Strategy::MaximumRootToLeafPath;
    throw Error(ErrKind::ParseStrategy, "Unparseable strategy: " + s);
}

struct Taxonomy {  // the start-up state of taxa2agg (:104-109)
    std::vector<Taxon> taxons;
    TaxonTree tree;
    TaxonList by_id;
    explicit Taxonomy(std::vector<Taxon> t) : taxons(std::move(t)), tree(taxons), by_id(taxons) {}
    std::vector<std::optional<TaxonId>> snapping(bool ranked_only) const { return tree.snapping(by_id, ranked_only); }
};

inline std::unique_ptr<Aggregator> make_aggregator(const Taxonomy& tx, Method m, Strategy s, float factor,
                                                   std::ostream* err) {  // :111-140
    if (m == Method::RangeMinimumQuery && s == Strategy::MaximumRootToLeafPath)
        return std::make_unique<RTLCalculator>(tx.tree.root, tx.by_id);
    if (m == Method::RangeMinimumQuery && s == Strategy::LowestCommonAncestor)
        return std::make_unique<RmqLCACalculator>(tx.tree);
    if (m == Method::RangeMinimumQuery && s == Strategy::Hybrid) {
        if (err) *err << "Warning: this is a hybrid between LCA/MRTL, not LCA*/MRTL\n";
        return std::make_unique<RmqMixCalculator>(tx.tree, factor);
    }
    if (m == Method::Tree && s == Strategy::LowestCommonAncestor)
        return std::make_unique<TreeLCACalculator>(tx.tree.root, tx.by_id);
    if (m == Method::Tree && s == Strategy::Hybrid)
        return std::make_unique<TreeMixCalculator>(tx.tree.root, tx.by_id, factor);
    // Debug formatting of the enum variants (:134-138)
    throw Error(ErrKind::InvalidInvocation,
                "Invalid invocation: Tree and MaximumRootToLeafPath cannot be combined");
}

// One taxa2agg record (:161-180) given already-parsed items.  Returns the set of strings the
// reference could print for this record; throws Error/Panic where every order errors.
struct RecordOutcome {
    std::set<TaxonId> printed;       // possible outputs
    std::set<TaxonId> raw;           // the aggregates they are the snaps of
    std::set<TaxonId> unknown;       // possible UnknownTaxon aborts
    bool may_panic = false;
    bool empty = false;              // counts.is_empty() => literal "1"
};
inline RecordOutcome taxa2agg_record(const Aggregator& agg, const std::vector<std::optional<TaxonId>>& snapping,
                                     const std::vector<std::pair<TaxonId, float>>& items, float lower_bound) {
    RecordOutcome ro;
    std::vector<std::pair<TaxonId, float>> nz;
    for (auto& p : items) if (p.first != 0) nz.push_back(p);  // :169
    Counts counts = filter(count(nz), lower_bound);           // :169-170
    if (counts.empty()) { ro.empty = true; ro.printed.insert(1); return ro; }  // :174-175 literal "1"
    AggOutcome ao = agg.aggregate(counts);
    ro.unknown = ao.unknown;
    ro.may_panic = ao.may_panic;
    for (TaxonId a : ao.ok) {
        if (a >= snapping.size() || !snapping[a]) ro.may_panic = true;  // snapping[aggregate].unwrap() (:178)
        else { ro.printed.insert(*snapping[a]); ro.raw.insert(a); }
    }
    return ro;
}

// Full command for deterministic comparison with the product CLI: ties are resolved with the
// product's documented rule via `pick` (given the tie set, choose one).
inline void taxa2agg(std::istream& in, std::ostream& out, std::ostream& err, const Taxonomy& tx, bool scored,
                     bool ranked_only, Method method, Strategy strategy, float factor, float lower_bound,
                     const std::function<TaxonId(const std::set<TaxonId>&)>& pick) {
    std::vector<std::optional<TaxonId>> snapping = tx.snapping(ranked_only);
    std::unique_ptr<Aggregator> agg = make_aggregator(tx, method, strategy, factor, &err);
    FastaWriter writer(out, "\n", false);
    FastaReader reader(in, false);
    Record record;
    while (reader.read_record(record)) {
        std::vector<std::pair<TaxonId, float>> items;
        for (auto& s : record.sequence) {
            if (scored) {  // with_score :142-148
                std::vector<std::string> split;
                size_t st = 0;
                for (;;) {
                    size_t p = s.find('=', st);
                    if (p == std::string::npos) { split.push_back(s.substr(st)); break; }
                    split.push_back(s.substr(st, p - st));
                    st = p + 1;
                }
                if (split.size() != 2) throw Error(ErrKind::Msg, "Taxon without score");
                TaxonId t = parse_usize(split[0]);
                float f = parse_f32(split[1]);
                items.emplace_back(t, f);
            } else {
                items.emplace_back(parse_usize(s), 1.0f);  // :150-152
            }
        }
        RecordOutcome ro = taxa2agg_record(*agg, snapping, items, lower_bound);
        if (!ro.unknown.empty()) throw unknown_taxon(pick(ro.unknown));
        if (ro.may_panic) throw Panic("index out of bounds / unwrap on None");
        Record o;
        o.header = record.header;
        // the product breaks a tie among the AGGREGATES (the nodes of the unsnapped tree) and snaps the winner -- not among the snaps
        o.sequence.push_back(std::to_string(ro.empty ? *ro.printed.begin() : *snapping[pick(ro.raw)]));
        writer.write_record(o);
    }
}

}  // namespace umgap_oracle
